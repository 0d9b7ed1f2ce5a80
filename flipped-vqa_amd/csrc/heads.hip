// Input/output heads of the Flipped-VQA step on gfx950: visual projection (+ temporal embedding),
// LM-head cross-entropy over fp32 logits, and the QAV frame-ordering head. All fp32 arithmetic,
// one wave (or workgroup) per row, fixed-order reductions (no float atomics ⇒ bitwise repeatable).
#include "common.h"

namespace {

// ---- visual projection: vf_raw[r,d] = video[r,:]·W[d,:]   (reference llama/model.py:322,324) --
// one wave per output feature d keeps W[d,:] in registers and sweeps the (few) frame rows
template <typename T, int KMAX>
__global__ __launch_bounds__(256) void visual_proj_fwd_k(const float* __restrict__ video, const float* __restrict__ W,
                                                         const float* __restrict__ temporal,
                                                         float* __restrict__ vf_raw, T* __restrict__ vf_tok, int R,
                                                         int F, int K, int D) {
  const int d = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (d >= D) return;
  const int lane = threadIdx.x & 63;
  float w[KMAX];
#pragma unroll
  for (int t = 0; t < KMAX / 4; ++t) {
    const int k = t * 256 + lane * 4;
    float v[4] = {0.f, 0.f, 0.f, 0.f};
    if (k < K) Vec4<float>::load(W + (size_t)d * K + k, v);
    w[4 * t] = v[0]; w[4 * t + 1] = v[1]; w[4 * t + 2] = v[2]; w[4 * t + 3] = v[3];
  }
  for (int r = 0; r < R; ++r) {
    float acc = 0.f;
#pragma unroll
    for (int t = 0; t < KMAX / 4; ++t) {
      const int k = t * 256 + lane * 4;
      if (k < K) {
        float v[4];
        Vec4<float>::load(video + (size_t)r * K + k, v);
        acc += v[0] * w[4 * t] + v[1] * w[4 * t + 1] + v[2] * w[4 * t + 2] + v[3] * w[4 * t + 3];
      }
    }
    acc = wave_sum(acc);
    if (lane == 0) {
      vf_raw[(size_t)r * D + d] = acc;
      vf_tok[(size_t)r * D + d] = from_f32<T>(acc + temporal[(size_t)(r % F) * D + d]);
    }
  }
}

// dW[d,k] += sum_r (d_tok[r,d] + d_qav[r,d]) * video[r,k]   (the weight gradient of llama/model.py:322).
// One wave owns DB output features and the whole K row of each in registers (DB x K/64 accumulators), so a frame row of
// `video` fetched from L2 feeds DB features (DB x fewer bytes through L2 than one feature per wave), four rows per trip.
template <int KMAX, int DB, int RT>   // RT frame rows per trip
__global__ __launch_bounds__(256) void visual_proj_bwd_k(const float* __restrict__ d_tok,
                                                         const float* __restrict__ d_qav,
                                                         const float* __restrict__ video, float* __restrict__ dW,
                                                         int R, int K, int D) {
  const int d0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * DB;
  if (d0 >= D) return;
  const int lane = threadIdx.x & 63;
  float acc[DB][KMAX];
#pragma unroll
  for (int j = 0; j < DB; ++j)
#pragma unroll
    for (int t = 0; t < KMAX; ++t) acc[j][t] = 0.f;
  for (int r0 = 0; r0 < R; r0 += RT) {                     // RT frame rows per trip: their loads are all in flight together (round 5: 8, was 4 —
                                                           // the kernel is a chain of R / RT load latencies: 40.6 us at 20 trips)
    float g[RT][DB];
    float v[RT][KMAX];
#pragma unroll
    for (int rr = 0; rr < RT; ++rr) {
      const int r = r0 + rr < R ? r0 + rr : R - 1;
      const float live = r0 + rr < R ? 1.f : 0.f;
#pragma unroll
      for (int j = 0; j < DB; ++j) {
        const int d = d0 + j < D ? d0 + j : D - 1;
        float gv = d_tok[(size_t)r * D + d];
        if (d_qav) gv += d_qav[(size_t)r * D + d];
        g[rr][j] = gv * live;
      }
#pragma unroll
      for (int t = 0; t < KMAX / 4; ++t) {
        const int k = t * 256 + lane * 4;
        float q[4] = {0.f, 0.f, 0.f, 0.f};
        if (k < K) Vec4<float>::load(video + (size_t)r * K + k, q);
        v[rr][4 * t] = q[0]; v[rr][4 * t + 1] = q[1]; v[rr][4 * t + 2] = q[2]; v[rr][4 * t + 3] = q[3];
      }
    }
#pragma unroll
    for (int rr = 0; rr < RT; ++rr)                        // rows in order: the same fma chain as a row-by-row loop
#pragma unroll
      for (int j = 0; j < DB; ++j)
#pragma unroll
        for (int t = 0; t < KMAX; ++t) acc[j][t] += g[rr][j] * v[rr][t];
  }
#pragma unroll
  for (int j = 0; j < DB; ++j) {
    if (d0 + j >= D) break;
#pragma unroll
    for (int t = 0; t < KMAX / 4; ++t) {
      const int k = t * 256 + lane * 4;
      if (k < K) {
        float v[4];
        Vec4<float>::load(dW + (size_t)(d0 + j) * K + k, v);
        v[0] += acc[j][4 * t]; v[1] += acc[j][4 * t + 1]; v[2] += acc[j][4 * t + 2]; v[3] += acc[j][4 * t + 3];
        Vec4<float>::store(dW + (size_t)(d0 + j) * K + k, v);
      }
    }
  }
}

// vf_raw[r,d] = video[r,:]·W[d,:] on the exact-fp32 matrix cores (v_mfma_f32_16x16x4_f32: the same fma chain per
// output as a scalar loop over k in 16-element strides), then vf_tok = cast(vf_raw + temporal[r % F]). One workgroup
// per 16 output features: its 4 waves split K, each keeps the R <= 128 frame rows as 16-row MFMA blocks; the 4
// partial sums meet in LDS. 256 workgroups instead of the 32 tiles of a 128x128 GEMM at M = 80: 69 -> ~8 us.
template <typename T>
__global__ __launch_bounds__(256) void visual_proj_fwd_mfma_k(const float* __restrict__ video,
                                                              const float* __restrict__ W,
                                                              const float* __restrict__ temporal,
                                                              float* __restrict__ vf_raw, T* __restrict__ vf_tok,
                                                              int R, int F, int K, int D) {
  __shared__ float part[4][8][16][17];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int li = lane & 15, g = lane >> 4;
  const int n0 = blockIdx.x * 16;
  const int nrb = (R + 15) >> 4;                           // 16-row blocks of frame rows (<= 8)
  const int kw = K / 4;                                    // this wave's K range (K % 64 == 0)
  int bn = n0 + li; bn = bn < D ? bn : D - 1;
  const float* bp = W + (size_t)bn * K + (size_t)w * kw + 4 * g;
  f32x4 acc[8];
#pragma unroll
  for (int b = 0; b < 8; ++b) acc[b] = f32x4{0.f, 0.f, 0.f, 0.f};
  // four k-steps of 16 per trip, every load of the trip in flight before its first MFMA (round 5: one step per trip was a chain of
  // twelve dependent load latencies, 28.9 us in the step); the accumulation order per output is unchanged (k ascending)
  constexpr int VU = 4;
  const float* ap[8];
#pragma unroll
  for (int b = 0; b < 8; ++b) {
    int ar = b * 16 + li; ar = ar < R ? ar : R - 1;
    ap[b] = video + (size_t)ar * K + (size_t)w * kw + 4 * g;
  }
  for (int k0 = 0; k0 < kw; k0 += 16 * VU) {
    float4 bf[VU], af[VU][8];
#pragma unroll
    for (int u = 0; u < VU; ++u) {
      const int k = k0 + 16 * u;
      const bool in = k < kw;
      bf[u] = in ? *reinterpret_cast<const float4*>(bp + k) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
      for (int b = 0; b < 8; ++b)
        af[u][b] = (in && b < nrb) ? *reinterpret_cast<const float4*>(ap[b] + k) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int u = 0; u < VU; ++u) {
      if (k0 + 16 * u >= kw) break;
#pragma unroll
      for (int b = 0; b < 8; ++b) {
        if (b >= nrb) break;
        acc[b] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[u][b].x, bf[u].x, acc[b], 0, 0, 0);     // D[row 4g+e][feature li]
        acc[b] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[u][b].y, bf[u].y, acc[b], 0, 0, 0);
        acc[b] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[u][b].z, bf[u].z, acc[b], 0, 0, 0);
        acc[b] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[u][b].w, bf[u].w, acc[b], 0, 0, 0);
      }
    }
  }
#pragma unroll
  for (int b = 0; b < 8; ++b)
#pragma unroll
    for (int e = 0; e < 4; ++e) part[w][b][4 * g + e][li] = acc[b][e];
  __syncthreads();
  for (int i = threadIdx.x; i < nrb * 256; i += 256) {
    const int b = i >> 8, rr = (i >> 4) & 15, c = i & 15;
    const int r = b * 16 + rr, d = n0 + c;
    if (r < R && d < D) {
      const float v = ((part[0][b][rr][c] + part[1][b][rr][c]) + part[2][b][rr][c]) + part[3][b][rr][c];
      vf_raw[(size_t)r * D + d] = v;
      vf_tok[(size_t)r * D + d] = from_f32<T>(v + temporal[(size_t)(r % F) * D + d]);
    }
  }
}

__global__ __launch_bounds__(256) void temporal_bwd_k(const float* __restrict__ d_tok, float* __restrict__ dtemp,
                                                      int B, int F, int D) {
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= F * D) return;
  float s = 0.f;
  for (int b = 0; b < B; ++b) s += d_tok[(size_t)b * F * D + idx];
  dtemp[idx] += s;
}

// ---- LM-head cross entropy (reference llama/model.py:349-350) --------------------------------
// one workgroup per (n, s) row with s < S-1 and a non-ignored label; other rows exit at once
// block-wide max / sum for blockDim.x == 1024 (16 waves); `red` is 16 floats of LDS
__device__ __forceinline__ float block_max_1024(float v, float* red) {
  v = wave_max(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  float m = red[0];
#pragma unroll
  for (int i = 1; i < 16; ++i) m = fmaxf(m, red[i]);
  return m;
}
__device__ __forceinline__ float block_sum_1024(float v, float* red) {
  v = wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  float s = red[0];
#pragma unroll
  for (int i = 1; i < 16; ++i) s += red[i];               // fixed order
  return s;
}

// 1024 threads per row; a row of up to 32768 logits is read ONCE, all of its loads in flight together (round 5: the scored rows are
// a few dozen workgroups, so a row's latency chain — two dependent passes of 31 loads per thread at 256 threads — was the kernel:
// 16.7 us); longer rows take the two-pass loop.
__global__ __launch_bounds__(1024) void ce_fwd_k(const float* __restrict__ logits, const int64_t* __restrict__ labels,
                                                 float* __restrict__ lse, float* __restrict__ rowloss, int S, int V,
                                                 int64_t ignore) {
  __shared__ float red[16];
  const int row = blockIdx.x;
  const int s = row % S;
  int64_t lab = ignore;
  if (s < S - 1) lab = labels[row + 1];
  if (lab == ignore || lab < 0 || lab >= V) {       // block-uniform
    if (threadIdx.x == 0) { rowloss[row] = 0.f; lse[row] = 0.f; }
    return;
  }
  const float* z = logits + (size_t)row * V;
  constexpr int MAXC = 8;
  float m = -INFINITY, sum = 0.f;
  if (V <= 1024 * 4 * MAXC) {
    float v[MAXC][4];
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
      const int c = (threadIdx.x + 1024 * i) * 4;
      if (c < V) Vec4<float>::load(z + c, v[i]);
      else v[i][0] = v[i][1] = v[i][2] = v[i][3] = -INFINITY;
    }
#pragma unroll
    for (int i = 0; i < MAXC; ++i) m = fmaxf(fmaxf(m, fmaxf(v[i][0], v[i][1])), fmaxf(v[i][2], v[i][3]));
    m = block_max_1024(m, red);
#pragma unroll
    for (int i = 0; i < MAXC; ++i) sum += __expf(v[i][0] - m) + __expf(v[i][1] - m) + __expf(v[i][2] - m) + __expf(v[i][3] - m);
  } else {
    for (int c = threadIdx.x * 4; c < V; c += 4096) {
      float v[4];
      Vec4<float>::load(z + c, v);
      m = fmaxf(fmaxf(m, fmaxf(v[0], v[1])), fmaxf(v[2], v[3]));
    }
    m = block_max_1024(m, red);
    for (int c = threadIdx.x * 4; c < V; c += 4096) {
      float v[4];
      Vec4<float>::load(z + c, v);
      sum += __expf(v[0] - m) + __expf(v[1] - m) + __expf(v[2] - m) + __expf(v[3] - m);
    }
  }
  sum = block_sum_1024(sum, red);
  if (threadIdx.x == 0) {
    const float l = m + logf(sum);
    lse[row] = l;
    rowloss[row] = l - z[lab];
  }
}

// fixed-order sum of row losses and count of scored rows -> loss_sum[0..1]
__global__ __launch_bounds__(256) void loss_reduce_k(const float* __restrict__ rowloss,
                                                     const int64_t* __restrict__ labels, float* __restrict__ out,
                                                     int rows, int S, int64_t ignore, int64_t nclass) {
  __shared__ float red[4];
  float a = 0.f, c = 0.f;
  for (int r = threadIdx.x; r < rows; r += 256) {
    const int s = r % S;
    if (s < S - 1) {
      const int64_t lab = labels[r + 1];
      if (lab != ignore && lab >= 0 && lab < nclass) { a += rowloss[r]; c += 1.f; }
    }
  }
  a = block_sum_256(a, red);
  c = block_sum_256(c, red);
  if (threadIdx.x == 0) { out[0] += a; out[1] += c; }
}

template <typename T>
__global__ __launch_bounds__(1024) void ce_bwd_k(const float* __restrict__ logits, const int64_t* __restrict__ labels,
                                                const float* __restrict__ lse, const float* __restrict__ loss_sum,
                                                const float* __restrict__ gscale, T* __restrict__ dlogits, int S,
                                                int V, int64_t ignore) {
  const int row = blockIdx.x;
  const int s = row % S;
  int64_t lab = ignore;
  if (s < S - 1) lab = labels[row + 1];
  T* dz = dlogits + (size_t)row * V;
  if (lab == ignore || lab < 0 || lab >= V) {
    const float zero[4] = {0.f, 0.f, 0.f, 0.f};
    for (int c = threadIdx.x * 4; c < V; c += 4096) Vec4<T>::store(dz + c, zero);
    return;
  }
  const float* z = logits + (size_t)row * V;
  const float l = lse[row];
  const float k = gscale[0] / loss_sum[1];
  for (int c = threadIdx.x * 4; c < V; c += 4096) {
    float v[4], o[4];
    Vec4<float>::load(z + c, v);
#pragma unroll
    for (int j = 0; j < 4; ++j) o[j] = (__expf(v[j] - l) - ((int64_t)(c + j) == lab ? 1.f : 0.f)) * k;
    Vec4<T>::store(dz + c, o);
  }
}

// ---- QAV head (reference llama/model.py:359-361) ----------------------------------------------
// one wave per (n, s) row; only rows whose label (at s+1) is a frame index do work
template <typename T, int FMAX>
__global__ __launch_bounds__(256) void qav_fwd_k(const T* __restrict__ xn, const float* __restrict__ vf,
                                                 const int64_t* __restrict__ labels, float* __restrict__ probs,
                                                 float* __restrict__ rowloss, int rows, int S, int D, int F,
                                                 float inv_tau) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int lane = threadIdx.x & 63;
  const int n = row / S, s = row % S;
  int64_t lab = -1;
  if (s < S - 1) lab = labels[row + 1];
  if (lab < 0 || lab >= F) {
    if (lane == 0) rowloss[row] = 0.f;
    return;
  }
  float z[FMAX];
#pragma unroll
  for (int f = 0; f < FMAX; ++f) z[f] = 0.f;
  const T* x = xn + (size_t)row * D;
  for (int c = lane * 4; c < D; c += 256) {
    float xv[4];
    Vec4<T>::load(x + c, xv);
#pragma unroll
    for (int f = 0; f < FMAX; ++f) {
      if (f < F) {
        float v[4];
        Vec4<float>::load(vf + ((size_t)n * F + f) * D + c, v);
        z[f] += xv[0] * v[0] + xv[1] * v[1] + xv[2] * v[2] + xv[3] * v[3];
      }
    }
  }
  float m = -INFINITY;
#pragma unroll
  for (int f = 0; f < FMAX; ++f) {
    z[f] = wave_sum(z[f]) * inv_tau;
    if (f < F) m = fmaxf(m, z[f]);
  }
  float sum = 0.f, zl = 0.f;
#pragma unroll
  for (int f = 0; f < FMAX; ++f)
    if (f < F) {
      sum += expf(z[f] - m);
      if (f == (int)lab) zl = z[f];
    }
  const float l = m + logf(sum);
  if (lane == 0) {
    rowloss[row] = l - zl;
#pragma unroll
    for (int f = 0; f < FMAX; ++f)
      if (f < F) probs[(size_t)row * F + f] = expf(z[f] - l);
  }
}

// dxn[row,:] = sum_f dl[f] * vf[n,f,:]  (zero on unscored rows)
template <typename T, int FMAX>
__global__ __launch_bounds__(256) void qav_bwd_x_k(const float* __restrict__ vf, const int64_t* __restrict__ labels,
                                                   const float* __restrict__ probs,
                                                   const float* __restrict__ loss_sum,
                                                   const float* __restrict__ gscale, T* __restrict__ dxn, int rows,
                                                   int S, int D, int F, float inv_tau) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int lane = threadIdx.x & 63;
  const int n = row / S, s = row % S;
  int64_t lab = -1;
  if (s < S - 1) lab = labels[row + 1];
  T* dx = dxn + (size_t)row * D;
  if (lab < 0 || lab >= F) {
    const float zero[4] = {0.f, 0.f, 0.f, 0.f};
    for (int c = lane * 4; c < D; c += 256) Vec4<T>::store(dx + c, zero);
    return;
  }
  const float k = gscale[0] / loss_sum[1] * inv_tau;
  float dl[FMAX];
#pragma unroll
  for (int f = 0; f < FMAX; ++f) dl[f] = f < F ? (probs[(size_t)row * F + f] - (f == (int)lab ? 1.f : 0.f)) * k : 0.f;
  for (int c = lane * 4; c < D; c += 256) {
    float o[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int f = 0; f < FMAX; ++f)
      if (f < F) {
        float v[4];
        Vec4<float>::load(vf + ((size_t)n * F + f) * D + c, v);
        o[0] += dl[f] * v[0]; o[1] += dl[f] * v[1]; o[2] += dl[f] * v[2]; o[3] += dl[f] * v[3];
      }
    Vec4<T>::store(dx + c, o);
  }
}

// d_raw[n,f,:] += sum_s dl[n,s,f] * xn[n,s,:]   — one workgroup per (n, f)
template <typename T>
__global__ __launch_bounds__(256) void qav_bwd_v_k(const T* __restrict__ xn, const int64_t* __restrict__ labels,
                                                   const float* __restrict__ probs,
                                                   const float* __restrict__ loss_sum,
                                                   const float* __restrict__ gscale, float* __restrict__ d_raw, int S,
                                                   int D, int F, float inv_tau) {
  const int n = blockIdx.x / F, f = blockIdx.x % F;
  const float k = gscale[0] / loss_sum[1] * inv_tau;
  for (int c = threadIdx.x * 4; c < D; c += 1024) {
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    for (int s = 0; s < S - 1; ++s) {
      const int64_t lab = labels[(size_t)n * S + s + 1];
      if (lab < 0 || lab >= F) continue;
      const size_t row = (size_t)n * S + s;
      const float dl = (probs[row * F + f] - (f == (int)lab ? 1.f : 0.f)) * k;
      float xv[4];
      Vec4<T>::load(xn + row * D + c, xv);
      acc[0] += dl * xv[0]; acc[1] += dl * xv[1]; acc[2] += dl * xv[2]; acc[3] += dl * xv[3];
    }
    float o[4];
    float* dst = d_raw + ((size_t)n * F + f) * D + c;
    Vec4<float>::load(dst, o);
    o[0] += acc[0]; o[1] += acc[1]; o[2] += acc[2]; o[3] += acc[3];
    Vec4<float>::store(dst, o);
  }
}

}  // namespace

extern "C" int fvqa_visual_proj_fwd(const float* video, const float* W, const float* temporal, float* vf_raw,
                                    void* vf_tok, int n_frames_total, int max_feats, int in_dim, int dim, int dtype,
                                    void* stream) {
  if (!video || !W || !temporal || !vf_raw || !vf_tok) return FVQA_EINVAL;
  if (!fvqa_dtype_ok(dtype)) return FVQA_EINVAL;
  if (n_frames_total <= 0 || max_feats <= 0 || in_dim <= 0 || in_dim % 4 || in_dim > 2048 || dim <= 0)
    return FVQA_ESHAPE;
  dim3 grid((dim + 3) / 4), block(256);
  hipStream_t st = (hipStream_t)stream;
  // The wave-per-feature kernel re-reads the whole frame matrix for every output feature (1 GB of L2 traffic at
  // D = 4096: 120 us). Shapes of the step (K % 64 == 0, at most 128 frame rows, 16-byte rows) run on the exact-fp32
  // matrix cores instead, temporal embedding and cast included.
  if (in_dim % 64 == 0 && n_frames_total <= 128 && !(((uintptr_t)video | (uintptr_t)W) & 15)) {
    const dim3 g16((dim + 15) / 16);
    if (dtype == FVQA_H16)
      hipLaunchKernelGGL(visual_proj_fwd_mfma_k<bf16_t>, g16, block, 0, st, video, W, temporal, vf_raw, (bf16_t*)vf_tok,
                         n_frames_total, max_feats, in_dim, dim);
    else
      hipLaunchKernelGGL(visual_proj_fwd_mfma_k<float>, g16, block, 0, st, video, W, temporal, vf_raw, (float*)vf_tok,
                         n_frames_total, max_feats, in_dim, dim);
    FVQA_CHECK_LAUNCH();
    return FVQA_OK;
  }
  if (dtype == FVQA_H16)
    hipLaunchKernelGGL((visual_proj_fwd_k<bf16_t, 32>), grid, block, 0, st, video, W, temporal, vf_raw,
                       (bf16_t*)vf_tok, n_frames_total, max_feats, in_dim, dim);
  else
    hipLaunchKernelGGL((visual_proj_fwd_k<float, 32>), grid, block, 0, st, video, W, temporal, vf_raw,
                       (float*)vf_tok, n_frames_total, max_feats, in_dim, dim);
  FVQA_CHECK_LAUNCH();
  return FVQA_OK;
}

extern "C" int fvqa_visual_proj_bwd(const float* d_tok, const float* d_qav, const float* video, float* dW,
                                    float* dtemporal, int n_frames_total, int max_feats, int in_dim, int dim,
                                    void* stream) {
  if (!d_tok || !video || !dW || !dtemporal) return FVQA_EINVAL;
  if (n_frames_total <= 0 || max_feats <= 0 || n_frames_total % max_feats || in_dim <= 0 || in_dim % 4 ||
      in_dim > 2048 || dim <= 0)
    return FVQA_ESHAPE;
  hipStream_t st = (hipStream_t)stream;
  if (in_dim <= 1024)        // 4 features per wave (4 x 16 accumulators), 4 waves per CU at D = 4096
    hipLaunchKernelGGL((visual_proj_bwd_k<16, 4, 8>), dim3((dim + 15) / 16), dim3(256), 0, st, d_tok, d_qav, video, dW,
                       n_frames_total, in_dim, dim);
  else
    hipLaunchKernelGGL((visual_proj_bwd_k<32, 2, 4>), dim3((dim + 7) / 8), dim3(256), 0, st, d_tok, d_qav, video, dW,
                       n_frames_total, in_dim, dim);
  hipLaunchKernelGGL(temporal_bwd_k, dim3((max_feats * dim + 255) / 256), dim3(256), 0, st, d_tok, dtemporal,
                     n_frames_total / max_feats, max_feats, dim);
  FVQA_CHECK_LAUNCH();
  return FVQA_OK;
}

extern "C" int fvqa_ce_fwd(const float* logits, const int64_t* labels, float* lse, float* rowloss, float* loss_sum,
                           int n_seq, int seq_len, int vocab, int64_t ignore_index, void* stream) {
  if (!logits || !labels || !lse || !rowloss || !loss_sum) return FVQA_EINVAL;
  if (n_seq <= 0 || seq_len < 2 || vocab <= 0 || vocab % 4) return FVQA_ESHAPE;
  hipStream_t st = (hipStream_t)stream;
  const int rows = n_seq * seq_len;
  hipLaunchKernelGGL(ce_fwd_k, dim3(rows), dim3(1024), 0, st, logits, labels, lse, rowloss, seq_len, vocab,
                     ignore_index);
  hipLaunchKernelGGL(loss_reduce_k, dim3(1), dim3(256), 0, st, rowloss, labels, loss_sum, rows, seq_len,
                     ignore_index, (int64_t)vocab);
  FVQA_CHECK_LAUNCH();
  return FVQA_OK;
}

extern "C" int fvqa_ce_bwd(const float* logits, const int64_t* labels, const float* lse, const float* loss_sum,
                           const float* gscale, void* dlogits, int n_seq, int seq_len, int vocab,
                           int64_t ignore_index, int dtype, void* stream) {
  if (!logits || !labels || !lse || !loss_sum || !gscale || !dlogits) return FVQA_EINVAL;
  if (!fvqa_dtype_ok(dtype)) return FVQA_EINVAL;
  if (n_seq <= 0 || seq_len < 2 || vocab <= 0 || vocab % 4) return FVQA_ESHAPE;
  hipStream_t st = (hipStream_t)stream;
  const int rows = n_seq * seq_len;
  if (dtype == FVQA_H16)
    hipLaunchKernelGGL(ce_bwd_k<bf16_t>, dim3(rows), dim3(1024), 0, st, logits, labels, lse, loss_sum, gscale,
                       (bf16_t*)dlogits, seq_len, vocab, ignore_index);
  else
    hipLaunchKernelGGL(ce_bwd_k<float>, dim3(rows), dim3(1024), 0, st, logits, labels, lse, loss_sum, gscale,
                       (float*)dlogits, seq_len, vocab, ignore_index);
  FVQA_CHECK_LAUNCH();
  return FVQA_OK;
}

extern "C" int fvqa_qav_head_fwd(const void* xn, const float* vf_raw, const int64_t* labels, float* probs,
                                 float* rowloss, float* loss_sum, int n_seq, int seq_len, int dim, int max_feats,
                                 float tau, int dtype, void* stream) {
  if (!xn || !vf_raw || !labels || !probs || !rowloss || !loss_sum) return FVQA_EINVAL;
  if (!fvqa_dtype_ok(dtype)) return FVQA_EINVAL;
  if (n_seq <= 0 || seq_len < 2 || dim <= 0 || dim % 4 || max_feats <= 0 || max_feats > 16 || tau == 0.f)
    return FVQA_ESHAPE;
  hipStream_t st = (hipStream_t)stream;
  const int rows = n_seq * seq_len;
  dim3 grid((rows + 3) / 4), block(256);
  if (dtype == FVQA_H16)
    hipLaunchKernelGGL((qav_fwd_k<bf16_t, 16>), grid, block, 0, st, (const bf16_t*)xn, vf_raw, labels, probs, rowloss,
                       rows, seq_len, dim, max_feats, 1.f / tau);
  else
    hipLaunchKernelGGL((qav_fwd_k<float, 16>), grid, block, 0, st, (const float*)xn, vf_raw, labels, probs, rowloss,
                       rows, seq_len, dim, max_feats, 1.f / tau);
  hipLaunchKernelGGL(loss_reduce_k, dim3(1), dim3(256), 0, st, rowloss, labels, loss_sum, rows, seq_len,
                     (int64_t)-1, (int64_t)max_feats);
  FVQA_CHECK_LAUNCH();
  return FVQA_OK;
}

extern "C" int fvqa_qav_head_bwd(const void* xn, const float* vf_raw, const int64_t* labels, const float* probs,
                                 const float* loss_sum, const float* gscale, void* dxn, float* d_raw, int n_seq,
                                 int seq_len, int dim, int max_feats, float tau, int dtype, void* stream) {
  if (!xn || !vf_raw || !labels || !probs || !loss_sum || !gscale || !dxn || !d_raw) return FVQA_EINVAL;
  if (!fvqa_dtype_ok(dtype)) return FVQA_EINVAL;
  if (n_seq <= 0 || seq_len < 2 || dim <= 0 || dim % 4 || max_feats <= 0 || max_feats > 16 || tau == 0.f)
    return FVQA_ESHAPE;
  hipStream_t st = (hipStream_t)stream;
  const int rows = n_seq * seq_len;
  dim3 grid((rows + 3) / 4), block(256);
  if (dtype == FVQA_H16) {
    hipLaunchKernelGGL((qav_bwd_x_k<bf16_t, 16>), grid, block, 0, st, vf_raw, labels, probs, loss_sum, gscale,
                       (bf16_t*)dxn, rows, seq_len, dim, max_feats, 1.f / tau);
    hipLaunchKernelGGL(qav_bwd_v_k<bf16_t>, dim3(n_seq * max_feats), block, 0, st, (const bf16_t*)xn, labels, probs,
                       loss_sum, gscale, d_raw, seq_len, dim, max_feats, 1.f / tau);
  } else {
    hipLaunchKernelGGL((qav_bwd_x_k<float, 16>), grid, block, 0, st, vf_raw, labels, probs, loss_sum, gscale,
                       (float*)dxn, rows, seq_len, dim, max_feats, 1.f / tau);
    hipLaunchKernelGGL(qav_bwd_v_k<float>, dim3(n_seq * max_feats), block, 0, st, (const float*)xn, labels, probs,
                       loss_sum, gscale, d_raw, seq_len, dim, max_feats, 1.f / tau);
  }
  FVQA_CHECK_LAUNCH();
  return FVQA_OK;
}
