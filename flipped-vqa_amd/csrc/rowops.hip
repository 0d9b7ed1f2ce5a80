// HBM-bound row kernels of the Flipped-VQA step: RMSNorm, RoPE, SwiGLU, embedding gather +
// frame splice. One 64-lane wave owns one activation row; every access is a 16-byte (fp32) or
// 8-byte (bf16) per-lane vector, rows are reduced with wavefront shuffles (no LDS, no atomics).
#include "common.h"

namespace {

constexpr int ROWS_PER_BLOCK = 4;   // 4 waves per 256-thread workgroup

// ---- RMSNorm (reference llama/model.py:37-42) ---------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void rmsnorm_fwd_k(const T* __restrict__ x, const T* __restrict__ w,
                                                     T* __restrict__ y, float* __restrict__ rstd, int rows,
                                                     int dim, float eps) {
  const int row = blockIdx.x * ROWS_PER_BLOCK + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int lane = threadIdx.x & 63;
  const T* xr = x + (size_t)row * dim;
  float ss = 0.f;
  for (int c = lane * 4; c < dim; c += 256) {
    float v[4];
    Vec4<T>::load(xr + c, v);
    ss += v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
  }
  ss = wave_sum(ss);
  const float r = rsqrtf(ss / (float)dim + eps);
  if (lane == 0 && rstd) rstd[row] = r;
  T* yr = y + (size_t)row * dim;
  for (int c = lane * 4; c < dim; c += 256) {
    float v[4], g[4], o[4];
    Vec4<T>::load(xr + c, v);
    Vec4<T>::load(w + c, g);
#pragma unroll
    for (int i = 0; i < 4; ++i) o[i] = round_to<T>(v[i] * r) * g[i];   // .type_as(x) then * weight
    Vec4<T>::store(yr + c, o);
  }
}

template <typename T>
__global__ __launch_bounds__(256) void rmsnorm_bwd_k(const T* __restrict__ g, const T* __restrict__ x,
                                                     const T* __restrict__ w, const float* __restrict__ rstd,
                                                     const T* __restrict__ resid, T* __restrict__ dx, int rows,
                                                     int dim) {
  const int row = blockIdx.x * ROWS_PER_BLOCK + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int lane = threadIdx.x & 63;
  const T* xr = x + (size_t)row * dim;
  const T* gr = g + (size_t)row * dim;
  float dot = 0.f;
  for (int c = lane * 4; c < dim; c += 256) {
    float xv[4], gv[4], wv[4];
    Vec4<T>::load(xr + c, xv);
    Vec4<T>::load(gr + c, gv);
    Vec4<T>::load(w + c, wv);
#pragma unroll
    for (int i = 0; i < 4; ++i) dot += gv[i] * wv[i] * xv[i];
  }
  dot = wave_sum(dot);
  const float r = rstd[row];
  const float k = r * r * r * dot / (float)dim;
  const T* rr = resid ? resid + (size_t)row * dim : nullptr;
  T* dr = dx + (size_t)row * dim;
  for (int c = lane * 4; c < dim; c += 256) {
    float xv[4], gv[4], wv[4], o[4];
    Vec4<T>::load(xr + c, xv);
    Vec4<T>::load(gr + c, gv);
    Vec4<T>::load(w + c, wv);
#pragma unroll
    for (int i = 0; i < 4; ++i) o[i] = r * gv[i] * wv[i] - xv[i] * k;
    if (rr) {
      float rv[4];
      Vec4<T>::load(rr + c, rv);
#pragma unroll
      for (int i = 0; i < 4; ++i) o[i] += rv[i];
    }
    Vec4<T>::store(dr + c, o);
  }
}

// ---- RoPE on q,k of the fused qkv rows (reference llama/model.py:61-67) ---------------------
template <typename T>
__global__ __launch_bounds__(256) void rope_qk_k(T* __restrict__ qkv, const float* __restrict__ cs,
                                                 const float* __restrict__ sn, int rows, int seq_len, int dim,
                                                 int head_dim, float sign) {
  const int row = blockIdx.x * ROWS_PER_BLOCK + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int lane = threadIdx.x & 63;
  const int pos = row % seq_len;
  const int hp = head_dim >> 1;
  T* base = qkv + (size_t)row * 3 * dim;
  for (int c = lane * 4; c < 2 * dim; c += 256) {      // q block then k block are contiguous
    const int d = (c % dim) % head_dim;                 // even, multiple of 4
    const int i0 = d >> 1;
    float v[4], o[4];
    Vec4<T>::load(base + c, v);
    const float c0 = cs[pos * hp + i0], s0 = sign * sn[pos * hp + i0];
    const float c1 = cs[pos * hp + i0 + 1], s1 = sign * sn[pos * hp + i0 + 1];
    o[0] = v[0] * c0 - v[1] * s0;
    o[1] = v[0] * s0 + v[1] * c0;
    o[2] = v[2] * c1 - v[3] * s1;
    o[3] = v[2] * s1 + v[3] * c1;
    Vec4<T>::store(base + c, o);
  }
}

// ---- SwiGLU (reference llama/model.py:142) --------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void swiglu_fwd_k(const T* __restrict__ ab, T* __restrict__ z, size_t n4,
                                                    int hidden) {
  const int h4 = hidden >> 2;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
    const size_t r = i / h4;
    const int c = (int)(i % h4) * 4;
    float a[4], b[4], o[4];
    Vec4<T>::load(ab + r * 2 * hidden + c, a);
    Vec4<T>::load(ab + r * 2 * hidden + hidden + c, b);
#pragma unroll
    for (int j = 0; j < 4; ++j) o[j] = round_to<T>(a[j] / (1.f + __expf(-a[j]))) * b[j];
    Vec4<T>::store(z + r * hidden + c, o);
  }
}

template <typename T>
__global__ __launch_bounds__(256) void swiglu_bwd_k(const T* __restrict__ dz, const T* __restrict__ ab,
                                                    T* __restrict__ dab, size_t n4, int hidden) {
  const int h4 = hidden >> 2;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
    const size_t r = i / h4;
    const int c = (int)(i % h4) * 4;
    float a[4], b[4], g[4], da[4], db[4];
    Vec4<T>::load(ab + r * 2 * hidden + c, a);
    Vec4<T>::load(ab + r * 2 * hidden + hidden + c, b);
    Vec4<T>::load(dz + r * hidden + c, g);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float sg = 1.f / (1.f + __expf(-a[j]));
      da[j] = g[j] * b[j] * sg * (1.f + a[j] * (1.f - sg));
      db[j] = g[j] * a[j] * sg;
    }
    Vec4<T>::store(dab + r * 2 * hidden + c, da);
    Vec4<T>::store(dab + r * 2 * hidden + hidden + c, db);
  }
}

// ---- embedding gather + frame splice (reference llama/model.py:286-294,326-336) -------------
template <typename T>
__global__ __launch_bounds__(256) void embed_splice_k(const int64_t* __restrict__ ids, const T* __restrict__ emb,
                                                      const T* __restrict__ vf, const int64_t* __restrict__ zlabels,
                                                      const int64_t* __restrict__ index, T* __restrict__ h,
                                                      int rows, int seq_len, int dim, int F, int vstart, int mode) {
  const int row = blockIdx.x * ROWS_PER_BLOCK + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int lane = threadIdx.x & 63;
  const int n = row / seq_len, s = row % seq_len;
  const T* src = emb + (size_t)ids[row] * dim;
  T* dst = h + (size_t)row * dim;
  if (mode == 0) {
    if (s >= vstart && s < vstart + F) src = vf + ((size_t)n * F + (s - vstart)) * dim;
    for (int c = lane * 4; c < dim; c += 256) {
      float v[4];
      Vec4<T>::load(src + c, v);
      Vec4<T>::store(dst + c, v);
    }
    return;
  }
  const bool zero = zlabels && zlabels[row] >= 0;   // qav_video_mask = qav_label.ge(0), model.py:282
  for (int c = lane * 4; c < dim; c += 256) {
    float v[4] = {0.f, 0.f, 0.f, 0.f};
    if (!zero) Vec4<T>::load(src + c, v);
    for (int f = 0; f < F; ++f) {                      // scatter_add_: every frame aimed at this row
      if (index[n * F + f] == (int64_t)s) {
        float u[4];
        Vec4<T>::load(vf + ((size_t)n * F + f) * dim + c, u);
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = round_to<T>(v[j] + u[j]);
      }
    }
    Vec4<T>::store(dst + c, v);
  }
}

template <typename T>
__global__ __launch_bounds__(256) void splice_bwd_k(const T* __restrict__ dh, const int64_t* __restrict__ index,
                                                    float* __restrict__ d_tok, int n_seq, int seq_len, int dim,
                                                    int F, int vstart, int mode) {
  const int nf = blockIdx.x;             // one workgroup per (sequence, frame)
  const int n = nf / F, f = nf % F;
  int64_t s = mode == 0 ? (int64_t)(vstart + f) : index[nf];
  if (s < 0 || s >= seq_len) return;
  const T* src = dh + ((size_t)n * seq_len + s) * dim;
  float* dst = d_tok + (size_t)nf * dim;
  for (int c = threadIdx.x * 4; c < dim; c += 1024) {
    float v[4], o[4];
    Vec4<T>::load(src + c, v);
    Vec4<float>::load(dst + c, o);
#pragma unroll
    for (int j = 0; j < 4; ++j) o[j] += v[j];
    Vec4<float>::store(dst + c, o);
  }
}

template <typename T>
__global__ __launch_bounds__(256) void cast_rows_k(const float* __restrict__ src, T* __restrict__ dst, size_t n4) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
    float v[4];
    Vec4<float>::load(src + i * 4, v);
    Vec4<T>::store(dst + i * 4, v);
  }
}

inline int grid_for(size_t n_items) {
  size_t g = (n_items + 255) / 256;
  return (int)(g < 1 ? 1 : (g > 4096 ? 4096 : g));
}

}  // namespace

#define DISPATCH_T(dtype, ...)                         \
  if ((dtype) == FVQA_BF16) { typedef bf16_t T; __VA_ARGS__; } \
  else { typedef float T; __VA_ARGS__; }

extern "C" int fvqa_rmsnorm_fwd(const void* x, const void* w, void* y, float* rstd, int rows, int dim, float eps,
                                int dtype, void* stream) {
  if (!x || !w || !y) return FVQA_EINVAL;
  if (!fvqa_dtype_ok(dtype)) return FVQA_EINVAL;
  if (rows <= 0 || dim <= 0 || dim % 4) return FVQA_ESHAPE;
  dim3 grid((rows + ROWS_PER_BLOCK - 1) / ROWS_PER_BLOCK), block(256);
  DISPATCH_T(dtype, hipLaunchKernelGGL(rmsnorm_fwd_k<T>, grid, block, 0, (hipStream_t)stream, (const T*)x,
                                       (const T*)w, (T*)y, rstd, rows, dim, eps));
  FVQA_CHECK_LAUNCH();
  return FVQA_OK;
}

extern "C" int fvqa_rmsnorm_bwd(const void* g, const void* x, const void* w, const float* rstd, const void* resid,
                                void* dx, int rows, int dim, int dtype, void* stream) {
  if (!g || !x || !w || !rstd || !dx) return FVQA_EINVAL;
  if (!fvqa_dtype_ok(dtype)) return FVQA_EINVAL;
  if (rows <= 0 || dim <= 0 || dim % 4) return FVQA_ESHAPE;
  dim3 grid((rows + ROWS_PER_BLOCK - 1) / ROWS_PER_BLOCK), block(256);
  DISPATCH_T(dtype, hipLaunchKernelGGL(rmsnorm_bwd_k<T>, grid, block, 0, (hipStream_t)stream, (const T*)g,
                                       (const T*)x, (const T*)w, rstd, (const T*)resid, (T*)dx, rows, dim));
  FVQA_CHECK_LAUNCH();
  return FVQA_OK;
}

extern "C" int fvqa_rope_qk(void* qkv, const float* cos_t, const float* sin_t, int n_seq, int seq_len, int n_heads,
                            int head_dim, int inverse, int dtype, void* stream) {
  if (!qkv || !cos_t || !sin_t) return FVQA_EINVAL;
  if (!fvqa_dtype_ok(dtype)) return FVQA_EINVAL;
  if (n_seq <= 0 || seq_len <= 0 || n_heads <= 0 || head_dim <= 0 || head_dim % 4) return FVQA_ESHAPE;
  const int rows = n_seq * seq_len, dim = n_heads * head_dim;
  dim3 grid((rows + ROWS_PER_BLOCK - 1) / ROWS_PER_BLOCK), block(256);
  const float sign = inverse ? -1.f : 1.f;
  DISPATCH_T(dtype, hipLaunchKernelGGL(rope_qk_k<T>, grid, block, 0, (hipStream_t)stream, (T*)qkv, cos_t, sin_t,
                                       rows, seq_len, dim, head_dim, sign));
  FVQA_CHECK_LAUNCH();
  return FVQA_OK;
}

extern "C" int fvqa_swiglu_fwd(const void* ab, void* z, int rows, int hidden, int dtype, void* stream) {
  if (!ab || !z) return FVQA_EINVAL;
  if (!fvqa_dtype_ok(dtype)) return FVQA_EINVAL;
  if (rows <= 0 || hidden <= 0 || hidden % 4) return FVQA_ESHAPE;
  const size_t n4 = (size_t)rows * (hidden / 4);
  DISPATCH_T(dtype, hipLaunchKernelGGL(swiglu_fwd_k<T>, dim3(grid_for(n4)), dim3(256), 0, (hipStream_t)stream,
                                       (const T*)ab, (T*)z, n4, hidden));
  FVQA_CHECK_LAUNCH();
  return FVQA_OK;
}

extern "C" int fvqa_swiglu_bwd(const void* dz, const void* ab, void* dab, int rows, int hidden, int dtype,
                               void* stream) {
  if (!dz || !ab || !dab) return FVQA_EINVAL;
  if (!fvqa_dtype_ok(dtype)) return FVQA_EINVAL;
  if (rows <= 0 || hidden <= 0 || hidden % 4) return FVQA_ESHAPE;
  const size_t n4 = (size_t)rows * (hidden / 4);
  DISPATCH_T(dtype, hipLaunchKernelGGL(swiglu_bwd_k<T>, dim3(grid_for(n4)), dim3(256), 0, (hipStream_t)stream,
                                       (const T*)dz, (const T*)ab, (T*)dab, n4, hidden));
  FVQA_CHECK_LAUNCH();
  return FVQA_OK;
}

extern "C" int fvqa_embed_splice(const int64_t* ids, const void* emb, const void* vf_tok, const int64_t* zero_labels,
                                 const int64_t* index, void* h, int n_seq, int seq_len, int dim, int max_feats,
                                 int vstart, int mode, int dtype, void* stream) {
  if (!ids || !emb || !vf_tok || !h) return FVQA_EINVAL;
  if (!fvqa_dtype_ok(dtype) || (mode != 0 && mode != 1)) return FVQA_EINVAL;
  if (mode == 1 && !index) return FVQA_EINVAL;
  if (n_seq <= 0 || seq_len <= 0 || dim <= 0 || dim % 4 || max_feats < 0) return FVQA_ESHAPE;
  if (mode == 0 && (vstart < 0 || vstart + max_feats > seq_len)) return FVQA_ESHAPE;
  const int rows = n_seq * seq_len;
  dim3 grid((rows + ROWS_PER_BLOCK - 1) / ROWS_PER_BLOCK), block(256);
  DISPATCH_T(dtype, hipLaunchKernelGGL(embed_splice_k<T>, grid, block, 0, (hipStream_t)stream, ids, (const T*)emb,
                                       (const T*)vf_tok, zero_labels, index, (T*)h, rows, seq_len, dim, max_feats,
                                       vstart, mode));
  FVQA_CHECK_LAUNCH();
  return FVQA_OK;
}

extern "C" int fvqa_splice_bwd(const void* dh, const int64_t* index, float* d_tok, int n_seq, int seq_len, int dim,
                               int max_feats, int vstart, int mode, int dtype, void* stream) {
  if (!dh || !d_tok) return FVQA_EINVAL;
  if (!fvqa_dtype_ok(dtype) || (mode != 0 && mode != 1)) return FVQA_EINVAL;
  if (mode == 1 && !index) return FVQA_EINVAL;
  if (n_seq <= 0 || seq_len <= 0 || dim <= 0 || dim % 4 || max_feats <= 0) return FVQA_ESHAPE;
  if (mode == 0 && (vstart < 0 || vstart + max_feats > seq_len)) return FVQA_ESHAPE;
  DISPATCH_T(dtype, hipLaunchKernelGGL(splice_bwd_k<T>, dim3(n_seq * max_feats), dim3(256), 0, (hipStream_t)stream,
                                       (const T*)dh, index, d_tok, n_seq, seq_len, dim, max_feats, vstart, mode));
  FVQA_CHECK_LAUNCH();
  return FVQA_OK;
}

extern "C" int fvqa_cast_rows(const float* src, void* dst, int n_rows, int dim, int dtype, void* stream) {
  if (!src || !dst) return FVQA_EINVAL;
  if (!fvqa_dtype_ok(dtype)) return FVQA_EINVAL;
  if (n_rows <= 0 || dim <= 0 || dim % 4) return FVQA_ESHAPE;
  const size_t n4 = (size_t)n_rows * dim / 4;
  DISPATCH_T(dtype, hipLaunchKernelGGL(cast_rows_k<T>, dim3(grid_for(n4)), dim3(256), 0, (hipStream_t)stream, src,
                                       (T*)dst, n4));
  FVQA_CHECK_LAUNCH();
  return FVQA_OK;
}
