// HBM-bound row kernels of the Flipped-VQA step: RMSNorm, RoPE, SwiGLU, embedding gather +
// frame splice. One 64-lane wave owns one activation row; every access is a 16-byte (fp32) or
// 8-byte (bf16) per-lane vector, rows are reduced with wavefront shuffles (no LDS, no atomics).
#include "common.h"

namespace {

constexpr int ROWS_PER_BLOCK = 4;   // 4 waves per 256-thread workgroup

// ---- RMSNorm (reference llama/model.py:37-42) ---------------------------------------------
// One 256-thread workgroup per row; a thread owns 8-element chunks c = 8*tid + 2048*k, loaded once
// with 16-byte (bf16) / 2x16-byte (fp32) vector loads and kept in registers across the reduction
// (single pass over HBM).
constexpr int NORM_MAXK = 4;          // dim <= 8192

template <typename T> __device__ __forceinline__ void load8(const T* p, float (&v)[8]);
template <> __device__ __forceinline__ void load8<float>(const float* p, float (&v)[8]) {
  const float4 a = *reinterpret_cast<const float4*>(p), b = *reinterpret_cast<const float4*>(p + 4);
  v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
}
template <> __device__ __forceinline__ void load8<bf16_t>(const bf16_t* p, float (&v)[8]) {
  const uint4 t = *reinterpret_cast<const uint4*>(p);
  v[0] = h16_lo(t.x); v[1] = h16_hi(t.x);
  v[2] = h16_lo(t.y); v[3] = h16_hi(t.y);
  v[4] = h16_lo(t.z); v[5] = h16_hi(t.z);
  v[6] = h16_lo(t.w); v[7] = h16_hi(t.w);
}
template <typename T> __device__ __forceinline__ void store8(T* p, const float (&v)[8]);
template <> __device__ __forceinline__ void store8<float>(float* p, const float (&v)[8]) {
  *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
  *reinterpret_cast<float4*>(p + 4) = make_float4(v[4], v[5], v[6], v[7]);
}
template <> __device__ __forceinline__ void store8<bf16_t>(bf16_t* p, const float (&v)[8]) {
  uint4 t;
  t.x = (unsigned)f32_to_bf16_bits(v[0]) | ((unsigned)f32_to_bf16_bits(v[1]) << 16);
  t.y = (unsigned)f32_to_bf16_bits(v[2]) | ((unsigned)f32_to_bf16_bits(v[3]) << 16);
  t.z = (unsigned)f32_to_bf16_bits(v[4]) | ((unsigned)f32_to_bf16_bits(v[5]) << 16);
  t.w = (unsigned)f32_to_bf16_bits(v[6]) | ((unsigned)f32_to_bf16_bits(v[7]) << 16);
  *reinterpret_cast<uint4*>(p) = t;
}
template <typename T>
__global__ __launch_bounds__(256) void rmsnorm_fwd_k(const T* __restrict__ x, const T* __restrict__ w,
                                                     T* __restrict__ y, float* __restrict__ rstd, int dim, float eps) {
  __shared__ float red[4];
  const int row = blockIdx.x, tid = threadIdx.x;
  const size_t base = (size_t)row * dim;
  float xv[NORM_MAXK][8];
  float ss = 0.f;
#pragma unroll
  for (int k = 0; k < NORM_MAXK; ++k) {
    const int c = tid * 8 + k * 2048;
    if (c < dim) {
      load8<T>(x + base + c, xv[k]);
#pragma unroll
      for (int i = 0; i < 8; ++i) ss += xv[k][i] * xv[k][i];
    }
  }
  ss = block_sum_256(ss, red);
  const float r = rsqrtf(ss / (float)dim + eps);
  if (tid == 0 && rstd) rstd[row] = r;
#pragma unroll
  for (int k = 0; k < NORM_MAXK; ++k) {
    const int c = tid * 8 + k * 2048;
    if (c < dim) {
      float g[8], o[8];
      load8<T>(w + c, g);
#pragma unroll
      for (int i = 0; i < 8; ++i) o[i] = round_to<T>(xv[k][i] * r) * g[i];   // .type_as(x) then * weight
      store8<T>(y + base + c, o);
    }
  }
}

template <typename T>
__global__ __launch_bounds__(256) void rmsnorm_bwd_k(const T* __restrict__ g, const T* __restrict__ x,
                                                     const T* __restrict__ w, const float* __restrict__ rstd,
                                                     const T* __restrict__ resid, T* __restrict__ dx, int dim) {
  __shared__ float red[4];
  const int row = blockIdx.x, tid = threadIdx.x;
  const size_t base = (size_t)row * dim;
  float xv[NORM_MAXK][8], gw[NORM_MAXK][8];
  float dot = 0.f;
#pragma unroll
  for (int k = 0; k < NORM_MAXK; ++k) {
    const int c = tid * 8 + k * 2048;
    if (c < dim) {
      float wv[8];
      load8<T>(x + base + c, xv[k]);
      load8<T>(g + base + c, gw[k]);
      load8<T>(w + c, wv);
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        gw[k][i] *= wv[i];
        dot += gw[k][i] * xv[k][i];
      }
    }
  }
  dot = block_sum_256(dot, red);
  const float r = rstd[row];
  const float kk = r * r * r * dot / (float)dim;
#pragma unroll
  for (int k = 0; k < NORM_MAXK; ++k) {
    const int c = tid * 8 + k * 2048;
    if (c < dim) {
      float o[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) o[i] = r * gw[k][i] - xv[k][i] * kk;
      if (resid) {
        float rv[8];
        load8<T>(resid + base + c, rv);
#pragma unroll
        for (int i = 0; i < 8; ++i) o[i] += rv[i];
      }
      store8<T>(dx + base + c, o);
    }
  }
}

// ---- RoPE on q,k of the fused qkv rows (reference llama/model.py:61-67) ---------------------
// one 256-thread workgroup per row, 8 elements (4 rotation pairs, 16 B of bf16) per thread
template <typename T>
__global__ __launch_bounds__(256) void rope_qk_k(T* __restrict__ qkv, const float* __restrict__ cs,
                                                 const float* __restrict__ sn, int seq_len, int dim,
                                                 int head_dim, float sign) {
  const int row = blockIdx.x;
  const int pos = row % seq_len;
  const int hp = head_dim >> 1;
  T* base = qkv + (size_t)row * 3 * dim;
  const float* cr = cs + (size_t)pos * hp;
  const float* sr = sn + (size_t)pos * hp;
  for (int c = threadIdx.x * 8; c < 2 * dim; c += 2048) {      // q block then k block are contiguous
    const int i0 = ((c % dim) % head_dim) >> 1;                 // first of 4 consecutive pair indices
    float v[8], o[8];
    load8<T>(base + c, v);
    const float4 c4 = *reinterpret_cast<const float4*>(cr + i0);
    const float4 s4 = *reinterpret_cast<const float4*>(sr + i0);
    const float cc[4] = {c4.x, c4.y, c4.z, c4.w}, ss[4] = {s4.x * sign, s4.y * sign, s4.z * sign, s4.w * sign};
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      o[2 * p] = v[2 * p] * cc[p] - v[2 * p + 1] * ss[p];
      o[2 * p + 1] = v[2 * p] * ss[p] + v[2 * p + 1] * cc[p];
    }
    store8<T>(base + c, o);
  }
}

// ---- SwiGLU (reference llama/model.py:142) --------------------------------------------------
// ab rows hold the W1 and W3 projections interleaved in 16-column blocks (include/fvqa.h "AB16"): column 32k + c is
// a[16k + c], column 32k + 16 + c is b[16k + c] (c < 16) — the layout in which one MFMA wave of the W1|W3 GEMM holds
// a and b of the same hidden unit in the same lane (gemm_sk.hip fuses this op into that GEMM's epilogue; these
// kernels serve the generation path and the tests).
__device__ __forceinline__ size_t ab16_col(int c) { return (size_t)(c >> 4) * 32 + (c & 15); }

template <typename T>
__global__ __launch_bounds__(256) void swiglu_fwd_k(const T* __restrict__ ab, T* __restrict__ z, size_t n4,
                                                    int hidden) {
  const int h4 = hidden >> 2;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
    const size_t r = i / h4;
    const int c = (int)(i % h4) * 4;
    float a[4], b[4], o[4];
    Vec4<T>::load(ab + r * 2 * hidden + ab16_col(c), a);
    Vec4<T>::load(ab + r * 2 * hidden + ab16_col(c) + 16, b);
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
    const size_t o = r * 2 * hidden + ab16_col(c);
    Vec4<T>::load(ab + o, a);
    Vec4<T>::load(ab + o + 16, b);
    Vec4<T>::load(dz + r * hidden + c, g);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float sg = 1.f / (1.f + __expf(-a[j]));
      da[j] = g[j] * b[j] * sg * (1.f + a[j] * (1.f - sg));
      db[j] = g[j] * a[j] * sg;
    }
    Vec4<T>::store(dab + o, da);
    Vec4<T>::store(dab + o + 16, db);
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

// Row movers of the tail rows (include/fvqa.h fvqa_row_segs): the rows a head reads, gathered from / scattered to up to three
// streams of the dense layout. 16-byte accesses; one workgroup per destination row; the segment table travels by value.
struct RowSegs {
  int n, stream_rows;
  int off[4];
  const int32_t* map[3];
};

template <typename T>
__global__ __launch_bounds__(256) void gather_rows_k(const T* __restrict__ src, T* __restrict__ dst, RowSegs sg, int dim) {
  constexpr int E = 16 / sizeof(T);
  const int j = blockIdx.x;
  int k = 0;
  while (k + 1 < sg.n && j >= sg.off[k + 1]) ++k;                         // (block-uniform; n <= 3)
  const int r = sg.map[k][j - sg.off[k]];
  const bool ok = r >= 0 && r < sg.stream_rows;                           // an index outside the stream reads as a zero row
  const uint4* s = reinterpret_cast<const uint4*>(src + ((size_t)k * sg.stream_rows + (ok ? r : 0)) * dim);
  uint4* d = reinterpret_cast<uint4*>(dst + (size_t)j * dim);
  for (int c = threadIdx.x; c < dim / E; c += 256) d[c] = ok ? s[c] : uint4{0u, 0u, 0u, 0u};
}

// dense row <- its compact row, or zeros (every dense row of the n streams is written exactly once: no atomics, no pre-clear)
template <typename T>
__global__ __launch_bounds__(256) void scatter_rows_k(const T* __restrict__ src, T* __restrict__ dst, RowSegs sg, int dim) {
  constexpr int E = 16 / sizeof(T);
  const int k = blockIdx.x / sg.stream_rows, r = blockIdx.x - k * sg.stream_rows;
  const int j = sg.map[k][r];
  const bool ok = j >= 0 && j < sg.off[k + 1] - sg.off[k];
  const uint4* s = reinterpret_cast<const uint4*>(src + (size_t)(sg.off[k] + (ok ? j : 0)) * dim);
  uint4* d = reinterpret_cast<uint4*>(dst + (size_t)blockIdx.x * dim);
  for (int c = threadIdx.x; c < dim / E; c += 256) d[c] = ok ? s[c] : uint4{0u, 0u, 0u, 0u};
}

inline int grid_for(size_t n_items) {
  size_t g = (n_items + 255) / 256;
  return (int)(g < 1 ? 1 : (g > 4096 ? 4096 : g));
}

}  // namespace

#define DISPATCH_T(dtype, ...)                         \
  if ((dtype) == FVQA_H16) { typedef bf16_t T; __VA_ARGS__; } \
  else { typedef float T; __VA_ARGS__; }

static inline int norm_dims_ok(int rows, int dim) {
  return rows > 0 && dim > 0 && dim % 8 == 0 && dim <= 2048 * NORM_MAXK;
}

extern "C" int fvqa_rmsnorm_fwd(const void* x, const void* w, void* y, float* rstd, int rows, int dim, float eps,
                                int dtype, void* stream) {
  if (!x || !w || !y) return FVQA_EINVAL;
  if (!fvqa_dtype_ok(dtype)) return FVQA_EINVAL;
  if (!norm_dims_ok(rows, dim)) return FVQA_ESHAPE;
  DISPATCH_T(dtype, hipLaunchKernelGGL((rmsnorm_fwd_k<T>), dim3(rows), dim3(256), 0, (hipStream_t)stream,
                                       (const T*)x, (const T*)w, (T*)y, rstd, dim, eps));
  FVQA_CHECK_LAUNCH();
  return FVQA_OK;
}

extern "C" int fvqa_rmsnorm_bwd(const void* g, const void* x, const void* w, const float* rstd, const void* resid,
                                void* dx, int rows, int dim, int dtype, void* stream) {
  if (!g || !x || !w || !rstd || !dx) return FVQA_EINVAL;
  if (!fvqa_dtype_ok(dtype)) return FVQA_EINVAL;
  if (!norm_dims_ok(rows, dim)) return FVQA_ESHAPE;
  DISPATCH_T(dtype, hipLaunchKernelGGL((rmsnorm_bwd_k<T>), dim3(rows), dim3(256), 0, (hipStream_t)stream,
                                       (const T*)g, (const T*)x, (const T*)w, rstd, (const T*)resid, (T*)dx, dim));
  FVQA_CHECK_LAUNCH();
  return FVQA_OK;
}

extern "C" int fvqa_rope_qk(void* qkv, const float* cos_t, const float* sin_t, int n_seq, int seq_len, int n_heads,
                            int head_dim, int inverse, int dtype, void* stream) {
  if (!qkv || !cos_t || !sin_t) return FVQA_EINVAL;
  if (!fvqa_dtype_ok(dtype)) return FVQA_EINVAL;
  if (n_seq <= 0 || seq_len <= 0 || n_heads <= 0 || head_dim <= 0 || head_dim % 4) return FVQA_ESHAPE;
  const int rows = n_seq * seq_len, dim = n_heads * head_dim;
  if (head_dim % 8) return FVQA_ESHAPE;
  const float sign = inverse ? -1.f : 1.f;
  DISPATCH_T(dtype, hipLaunchKernelGGL(rope_qk_k<T>, dim3(rows), dim3(256), 0, (hipStream_t)stream, (T*)qkv, cos_t,
                                       sin_t, seq_len, dim, head_dim, sign));
  FVQA_CHECK_LAUNCH();
  return FVQA_OK;
}

extern "C" int fvqa_swiglu_fwd(const void* ab, void* z, int rows, int hidden, int dtype, void* stream) {
  if (!ab || !z) return FVQA_EINVAL;
  if (!fvqa_dtype_ok(dtype)) return FVQA_EINVAL;
  if (rows <= 0 || hidden <= 0 || hidden % 16) return FVQA_ESHAPE;
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
  if (rows <= 0 || hidden <= 0 || hidden % 16) return FVQA_ESHAPE;
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

static inline int segs_ok(const fvqa_row_segs* g, RowSegs* out) {
  if (!g || g->n < 1 || g->n > 3 || g->stream_rows <= 0 || g->off[0] != 0) return 0;
  out->n = g->n; out->stream_rows = g->stream_rows;
  for (int k = 0; k < 4; ++k) out->off[k] = k <= g->n ? g->off[k] : g->off[g->n];
  for (int k = 0; k < 3; ++k) out->map[k] = k < g->n ? g->map[k] : nullptr;
  for (int k = 0; k < g->n; ++k)
    if (!g->map[k] || g->off[k + 1] < g->off[k]) return 0;
  return g->off[g->n] > 0;
}

extern "C" int fvqa_gather_rows(const void* src, void* dst, const fvqa_row_segs* segs, int dim, int dtype, void* stream) {
  if (!src || !dst) return FVQA_EINVAL;
  if (!fvqa_dtype_ok(dtype)) return FVQA_EINVAL;
  RowSegs sg;
  if (!segs_ok(segs, &sg)) return FVQA_EINVAL;
  const int e = 16 / (int)fvqa_dtype_size(dtype);
  if (dim <= 0 || dim % e) return FVQA_ESHAPE;
  if ((((uintptr_t)src | (uintptr_t)dst) & 15) != 0) return FVQA_EALIGN;
  DISPATCH_T(dtype, hipLaunchKernelGGL(gather_rows_k<T>, dim3(sg.off[sg.n]), dim3(256), 0, (hipStream_t)stream, (const T*)src,
                                       (T*)dst, sg, dim));
  FVQA_CHECK_LAUNCH();
  return FVQA_OK;
}

extern "C" int fvqa_scatter_rows(const void* src, void* dst, const fvqa_row_segs* segs, int dim, int dtype, void* stream) {
  if (!src || !dst) return FVQA_EINVAL;
  if (!fvqa_dtype_ok(dtype)) return FVQA_EINVAL;
  RowSegs sg;
  if (!segs_ok(segs, &sg)) return FVQA_EINVAL;
  const int e = 16 / (int)fvqa_dtype_size(dtype);
  if (dim <= 0 || dim % e) return FVQA_ESHAPE;
  if ((((uintptr_t)src | (uintptr_t)dst) & 15) != 0) return FVQA_EALIGN;
  DISPATCH_T(dtype, hipLaunchKernelGGL(scatter_rows_k<T>, dim3(sg.n * sg.stream_rows), dim3(256), 0, (hipStream_t)stream,
                                       (const T*)src, (T*)dst, sg, dim));
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
