// Adapter-gated prefix attention + causal attention, forward and backward, for gfx950.
//
// Restates reference llama/model.py:98-126 as ONE fused kernel per direction instead of the
// ~12 element-wise/softmax/cat/clone torch kernels that materialise (N,H,S,A+S) scores:
//   O = tanh(gate1[h]) · softmax(q·K_aᵀ/√Dh) · V_a  +  softmax(q·Kᵀ/√Dh + causal + gate2-bias) · V
// The two softmaxes are independent (rows sum to 1 + tanh(gate1)); gate2[h] is added on rows
// >= vs+F, cols [vs, vs+F) of sequences whose vstart >= 0 (VQA/VAQ streams only).
//
// Attention is ~1 % of the step's FLOPs at S=128, so it runs on the vector ALUs in fp32 (MFMA is
// kept for the projection GEMMs): K/V (or Q/dO) tiles of 64 rows are staged in LDS as fp32,
// a query (or key) row is owned by an aligned quad of lanes (32 of the 128 head dims each, in
// 16-byte interleaved chunks so a quad reads 64 contiguous LDS bytes and 16 quads broadcast),
// dot products finish with two wavefront shuffles, the row softmax is online over 16-key chunks.
// Backward = two kernels (dQ by query rows; dK/dV by key rows, with the batch-summed adapter
// key/value gradient as an extra key block) + a deterministic reduce; no float atomics.
#include "common.h"
#include <stdlib.h>

namespace {

constexpr int DH = 128;        // head dim (LLaMA 7B/13B/33B/65B)
constexpr int TILE = 64;       // rows staged per LDS tile
constexpr int CHUNK = 8;       // keys per online-softmax step
constexpr float NEG_BIG = -1e30f;

__device__ __forceinline__ float dot32(const float (&x)[32], const float* row, int part) {
  float a0 = 0.f, a1 = 0.f;
#pragma unroll
  for (int c = 0; c < 8; c += 2) {
    const float4 u = *reinterpret_cast<const float4*>(row + 16 * c + 4 * part);
    const float4 v = *reinterpret_cast<const float4*>(row + 16 * (c + 1) + 4 * part);
    a0 += x[4 * c] * u.x + x[4 * c + 1] * u.y + x[4 * c + 2] * u.z + x[4 * c + 3] * u.w;
    a1 += x[4 * c + 4] * v.x + x[4 * c + 5] * v.y + x[4 * c + 6] * v.z + x[4 * c + 7] * v.w;
  }
  return a0 + a1;
}
__device__ __forceinline__ void axpy32(float (&y)[32], float a, const float* row, int part) {
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    const float4 u = *reinterpret_cast<const float4*>(row + 16 * c + 4 * part);
    y[4 * c] += a * u.x; y[4 * c + 1] += a * u.y; y[4 * c + 2] += a * u.z; y[4 * c + 3] += a * u.w;
  }
}
template <typename T>
__device__ __forceinline__ void load_row32(const T* p, int part, float (&x)[32], float scale) {
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    float v[4];
    Vec4<T>::load(p + 16 * c + 4 * part, v);
    x[4 * c] = v[0] * scale; x[4 * c + 1] = v[1] * scale; x[4 * c + 2] = v[2] * scale; x[4 * c + 3] = v[3] * scale;
  }
}
template <typename T>
__device__ __forceinline__ void store_row32(T* p, int part, const float (&x)[32], float scale) {
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    float v[4] = {x[4 * c] * scale, x[4 * c + 1] * scale, x[4 * c + 2] * scale, x[4 * c + 3] * scale};
    Vec4<T>::store(p + 16 * c + 4 * part, v);
  }
}
// stage `nrows` rows x 128 dims (global row stride `ld`, rows clamped to last_row) into LDS fp32
template <typename T>
__device__ __forceinline__ void stage_tile(float* dst, const T* src0, size_t ld, int first_row, int last_row,
                                           int nrows, float scale) {
  for (int idx = threadIdx.x; idx < nrows * 32; idx += 256) {
    const int r = idx >> 5, c4 = idx & 31;
    int g = first_row + r;
    g = g < last_row ? g : last_row;
    float v[4];
    Vec4<T>::load(src0 + (size_t)g * ld + c4 * 4, v);
    *reinterpret_cast<float4*>(dst + r * DH + c4 * 4) = make_float4(v[0] * scale, v[1] * scale, v[2] * scale, v[3] * scale);
  }
}

// ------------------------------------------------------------------------------- forward
template <typename T>
__global__ __launch_bounds__(256, 2) void attn_fwd_k(const T* __restrict__ qkv, T* __restrict__ o,
                                                  float* __restrict__ lse_a, float* __restrict__ lse_t,
                                                  const float* __restrict__ gate1, const float* __restrict__ gate2,
                                                  const int32_t* __restrict__ vstart, int n_seq, int S, int H, int A,
                                                  int F) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float* sK = sm;                   // [TILE][DH]
  float* sV = sK + TILE * DH;       // [TILE][DH]
  float* sKa = sV + TILE * DH;      // [A][DH]
  float* sVa = sKa + A * DH;        // [A][DH]
  const int qb = blockIdx.x, h = blockIdx.y, n = blockIdx.z;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int part = lane & 3, rloc = lane >> 2;
  const int D = H * DH;
  const size_t ld = (size_t)3 * D;
  const int i = qb * TILE + w * 16 + rloc;          // query row of this quad
  const int ic = i < S ? i : S - 1;
  const float sc = rsqrtf((float)DH);
  const T* seq = qkv + (size_t)n * S * ld + h * DH;  // q block of this (n, h)
  float q[32], acc[32];
  load_row32<T>(seq + (size_t)ic * ld, part, q, sc);
#pragma unroll
  for (int c = 0; c < 32; ++c) acc[c] = 0.f;

  const int vs = vstart[n];
  const float g2 = gate2[h];
  const bool biased_row = vs >= 0 && ic >= vs + F;
  float m = NEG_BIG, l = 0.f;
  const int wave_last = min(qb * TILE + w * 16 + 15, S - 1);   // last query row this wave owns
  const int ntiles = qb + 1;
  for (int t = 0; t < ntiles; ++t) {
    __syncthreads();
    stage_tile<T>(sK, seq + D, ld, t * TILE, S - 1, TILE, 1.f);
    stage_tile<T>(sV, seq + 2 * D, ld, t * TILE, S - 1, TILE, 1.f);
    __syncthreads();
    for (int ch = 0; ch < TILE / CHUNK; ++ch) {
      const int j0 = t * TILE + ch * CHUNK;
      if (j0 > wave_last) break;                  // wave-uniform: whole chunk above the diagonal
      float s[CHUNK];
      float cmax = NEG_BIG;
#pragma unroll
      for (int jj = 0; jj < CHUNK; ++jj) {
        const int j = j0 + jj;
        float v = quad_sum(dot32(q, sK + (ch * CHUNK + jj) * DH, part));
        if (biased_row && j >= vs && j < vs + F) v += g2;
        v = (j <= ic) ? v : NEG_BIG;
        s[jj] = v;
        cmax = fmaxf(cmax, v);
      }
      const float mn = fmaxf(m, cmax);
      const float alpha = __expf(m - mn);
      l *= alpha;
#pragma unroll
      for (int c = 0; c < 32; ++c) acc[c] *= alpha;
      m = mn;
#pragma unroll
      for (int jj = 0; jj < CHUNK; ++jj) {
        const float p = (s[jj] > 0.5f * NEG_BIG) ? __expf(s[jj] - m) : 0.f;
        l += p;
        axpy32(acc, p, sV + (ch * CHUNK + jj) * DH, part);
      }
    }
  }
  const float inv_l = 1.f / l;
#pragma unroll
  for (int c = 0; c < 32; ++c) acc[c] *= inv_l;
  const float lt = m + __logf(l);

  // ---- adapter prefix: separate softmax over the A adapter keys, scaled by tanh(gate1[h])
  const T* arow = qkv + (size_t)n_seq * S * ld + h * DH;
  __syncthreads();
  stage_tile<T>(sKa, arow + D, ld, 0, A - 1, A, 1.f);
  stage_tile<T>(sVa, arow + 2 * D, ld, 0, A - 1, A, 1.f);
  __syncthreads();
  float ma = NEG_BIG, la = 0.f;
  for (int a = 0; a < A; ++a) {
    const float v = quad_sum(dot32(q, sKa + a * DH, part));
    const float mn = fmaxf(ma, v);
    la = la * __expf(ma - mn) + __expf(v - mn);
    ma = mn;
  }
  const float lsa = ma + __logf(la);
  const float g1 = tanhf(gate1[h]);
  for (int a = 0; a < A; ++a) {
    const float v = quad_sum(dot32(q, sKa + a * DH, part));
    axpy32(acc, g1 * __expf(v - lsa), sVa + a * DH, part);
  }
  if (i < S) {
    store_row32<T>(o + ((size_t)n * S + i) * D + h * DH, part, acc, 1.f);
    if (part == 0) {
      lse_a[((size_t)n * H + h) * S + i] = lsa;
      lse_t[((size_t)n * H + h) * S + i] = lt;
    }
  }
}

// ------------------------------------------------------------------------------- backward: dQ
// also emits delta_a = sum_a P_a dP_a, delta_t = dO·O - delta_a per row and the per-block
// partial sums of d tanh-gate (sum dO·V_a ⊙ P_a) and d gate2 (sum of dS_t over the bias window).
template <typename T>
__global__ __launch_bounds__(256) void attn_bwd_dq_k(const T* __restrict__ d_o, const T* __restrict__ qkv,
                                                     const T* __restrict__ o, const float* __restrict__ lse_a,
                                                     const float* __restrict__ lse_t,
                                                     const float* __restrict__ gate1,
                                                     const float* __restrict__ gate2,
                                                     const int32_t* __restrict__ vstart, T* __restrict__ dqkv,
                                                     float* __restrict__ delta_a, float* __restrict__ delta_t,
                                                     float* __restrict__ gate_part, int n_seq, int S, int H, int A,
                                                     int F) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float* sK = sm;
  float* sV = sK + TILE * DH;
  float* sKa = sV + TILE * DH;
  float* sVa = sKa + A * DH;
  __shared__ float red[8];
  const int qb = blockIdx.x, h = blockIdx.y, n = blockIdx.z;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int part = lane & 3, rloc = lane >> 2;
  const int D = H * DH;
  const size_t ld = (size_t)3 * D;
  const int i = qb * TILE + w * 16 + rloc;
  const bool live = i < S;
  const int ic = live ? i : S - 1;
  const float sc = rsqrtf((float)DH);
  const T* seq = qkv + (size_t)n * S * ld + h * DH;
  float q[32], dO[32], dq[32];
  load_row32<T>(seq + (size_t)ic * ld, part, q, sc);
  load_row32<T>(d_o + ((size_t)n * S + ic) * D + h * DH, part, dO, 1.f);
  float dtot;
  {
    float ov[32];
    load_row32<T>(o + ((size_t)n * S + ic) * D + h * DH, part, ov, 1.f);
    float t = 0.f;
#pragma unroll
    for (int c = 0; c < 32; ++c) t += ov[c] * dO[c];
    dtot = quad_sum(t);
  }
#pragma unroll
  for (int c = 0; c < 32; ++c) dq[c] = 0.f;
  const size_t sidx = ((size_t)n * H + h) * S + ic;
  const float lsa = lse_a[sidx], lst = lse_t[sidx];
  const float g1 = tanhf(gate1[h]);
  const float g2 = gate2[h];
  const int vs = vstart[n];
  const bool biased_row = vs >= 0 && ic >= vs + F;

  // adapter part
  const T* arow = qkv + (size_t)n_seq * S * ld + h * DH;
  stage_tile<T>(sKa, arow + D, ld, 0, A - 1, A, 1.f);
  stage_tile<T>(sVa, arow + 2 * D, ld, 0, A - 1, A, 1.f);
  __syncthreads();
  float da = 0.f, dg1 = 0.f;
  for (int a = 0; a < A; ++a) {
    const float p = __expf(quad_sum(dot32(q, sKa + a * DH, part)) - lsa);
    const float dov = quad_sum(dot32(dO, sVa + a * DH, part));
    dg1 += dov * p;
    da += p * g1 * dov;
  }
  for (int a = 0; a < A; ++a) {
    const float p = __expf(quad_sum(dot32(q, sKa + a * DH, part)) - lsa);
    const float dov = quad_sum(dot32(dO, sVa + a * DH, part));
    axpy32(dq, p * (g1 * dov - da), sKa + a * DH, part);
  }
  const float dt = dtot - da;
  float dg2 = 0.f;

  const int wave_last = min(qb * TILE + w * 16 + 15, S - 1);
  const int ntiles = qb + 1;
  for (int t = 0; t < ntiles; ++t) {
    __syncthreads();
    stage_tile<T>(sK, seq + D, ld, t * TILE, S - 1, TILE, 1.f);
    stage_tile<T>(sV, seq + 2 * D, ld, t * TILE, S - 1, TILE, 1.f);
    __syncthreads();
    const int jend = min(TILE, wave_last - t * TILE + 1);    // wave-uniform
    for (int jj = 0; jj < jend; ++jj) {
      const int j = t * TILE + jj;
      float s = quad_sum(dot32(q, sK + jj * DH, part));
      const bool inwin = biased_row && j >= vs && j < vs + F;
      if (inwin) s += g2;
      const float p = (j <= ic) ? __expf(s - lst) : 0.f;
      const float dp = quad_sum(dot32(dO, sV + jj * DH, part));
      const float ds = p * (dp - dt);
      if (inwin) dg2 += ds;
      axpy32(dq, ds, sK + jj * DH, part);
    }
  }
  if (live) {
    store_row32<T>(dqkv + ((size_t)n * S + i) * ld + h * DH, part, dq, sc);
    if (part == 0) {
      delta_a[sidx] = da;
      delta_t[sidx] = dt;
    }
  }
  // per-block gate partials (each row counted once: quad leader, live rows only)
  const float c1 = (live && part == 0) ? dg1 : 0.f;
  const float c2 = (live && part == 0) ? dg2 : 0.f;
  const float b1 = block_sum_256(c1, red);
  const float b2 = block_sum_256(c2, red + 4);
  if (threadIdx.x == 0) {
    const size_t pidx = (((size_t)n * H + h) * gridDim.x + qb) * 2;
    gate_part[pidx] = b1;
    gate_part[pidx + 1] = b2;
  }
}

// ------------------------------------------------------------------------------- backward: dK, dV
// blockIdx.x < nkb : 64 text keys (quad per key row, queries streamed through LDS)
// blockIdx.x == nkb: the A adapter keys; wave w takes queries i ≡ w (mod 4), partials meet in LDS.
template <typename T>
__global__ __launch_bounds__(256) void attn_bwd_dkv_k(const T* __restrict__ d_o, const T* __restrict__ qkv,
                                                      const float* __restrict__ lse_a,
                                                      const float* __restrict__ lse_t,
                                                      const float* __restrict__ delta_a,
                                                      const float* __restrict__ delta_t,
                                                      const float* __restrict__ gate1,
                                                      const float* __restrict__ gate2,
                                                      const int32_t* __restrict__ vstart, T* __restrict__ dqkv,
                                                      float* __restrict__ dka_part, float* __restrict__ dva_part,
                                                      int n_seq, int S, int H, int A, int F, int nkb) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float* sQ = sm;                    // [TILE][DH]  (pre-scaled by 1/sqrt(Dh))
  float* sdO = sQ + TILE * DH;       // [TILE][DH]
  float* sL = sdO + TILE * DH;       // [TILE] lse
  float* sDl = sL + TILE;            // [TILE] delta
  const int kb = blockIdx.x, h = blockIdx.y, n = blockIdx.z;
  const bool adapter = kb == nkb;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int part = lane & 3, rloc = lane >> 2;
  const int D = H * DH;
  const size_t ld = (size_t)3 * D;
  const float sc = rsqrtf((float)DH);
  const T* seq = qkv + (size_t)n * S * ld + h * DH;
  const size_t sbase = ((size_t)n * H + h) * S;

  float k[32], v[32], dk[32], dv[32];
#pragma unroll
  for (int c = 0; c < 32; ++c) { dk[c] = 0.f; dv[c] = 0.f; }

  if (!adapter) {
    const int j = kb * TILE + w * 16 + rloc;
    const int jc = j < S ? j : S - 1;
    load_row32<T>(seq + (size_t)jc * ld + D, part, k, 1.f);
    load_row32<T>(seq + (size_t)jc * ld + 2 * D, part, v, 1.f);
    const int vs = vstart[n];
    const float g2 = gate2[h];
    const bool biased_key = vs >= 0 && jc >= vs && jc < vs + F;
    const int wave_first = kb * TILE + w * 16;        // smallest key row of this wave
    const int nqt = (S + TILE - 1) / TILE;
    for (int t = kb; t < nqt; ++t) {
      __syncthreads();
      stage_tile<T>(sQ, seq, ld, t * TILE, S - 1, TILE, sc);
      stage_tile<T>(sdO, d_o + (size_t)n * S * D + h * DH, (size_t)D, t * TILE, S - 1, TILE, 1.f);
      if (threadIdx.x < TILE) {
        const int ii = min(t * TILE + (int)threadIdx.x, S - 1);
        sL[threadIdx.x] = lse_t[sbase + ii];
        sDl[threadIdx.x] = delta_t[sbase + ii];
      }
      __syncthreads();
      const int iend = min(TILE, S - t * TILE);
      const int ibeg = max(0, wave_first - t * TILE);    // wave-uniform: rows below every key of the wave
      for (int ii = ibeg; ii < iend; ++ii) {
        const int i = t * TILE + ii;
        float s = quad_sum(dot32(k, sQ + ii * DH, part));
        if (biased_key && i >= vs + F) s += g2;
        const float p = (jc <= i) ? __expf(s - sL[ii]) : 0.f;
        axpy32(dv, p, sdO + ii * DH, part);
        const float dp = quad_sum(dot32(v, sdO + ii * DH, part));
        axpy32(dk, p * (dp - sDl[ii]), sQ + ii * DH, part);
      }
    }
    if (j < S) {
      store_row32<T>(dqkv + ((size_t)n * S + j) * ld + h * DH + D, part, dk, 1.f);
      store_row32<T>(dqkv + ((size_t)n * S + j) * ld + h * DH + 2 * D, part, dv, 1.f);
    }
    return;
  }

  // ---- adapter keys
  const int a = rloc;                              // rows a >= A idle
  const int ac = a < A ? a : A - 1;
  const T* arow = qkv + (size_t)n_seq * S * ld + h * DH;
  load_row32<T>(arow + (size_t)ac * ld + D, part, k, 1.f);
  load_row32<T>(arow + (size_t)ac * ld + 2 * D, part, v, 1.f);
  const float g1 = tanhf(gate1[h]);
  const int nqt = (S + TILE - 1) / TILE;
  for (int t = 0; t < nqt; ++t) {
    __syncthreads();
    stage_tile<T>(sQ, seq, ld, t * TILE, S - 1, TILE, sc);
    stage_tile<T>(sdO, d_o + (size_t)n * S * D + h * DH, (size_t)D, t * TILE, S - 1, TILE, 1.f);
    if (threadIdx.x < TILE) {
      const int ii = min(t * TILE + (int)threadIdx.x, S - 1);
      sL[threadIdx.x] = lse_a[sbase + ii];
      sDl[threadIdx.x] = delta_a[sbase + ii];
    }
    __syncthreads();
    const int iend = min(TILE, S - t * TILE);
    for (int ii = w; ii < iend; ii += 4) {
      const float s = quad_sum(dot32(k, sQ + ii * DH, part));
      const float p = __expf(s - sL[ii]);
      axpy32(dv, g1 * p, sdO + ii * DH, part);
      const float dov = quad_sum(dot32(v, sdO + ii * DH, part));
      axpy32(dk, p * (g1 * dov - sDl[ii]), sQ + ii * DH, part);
    }
  }
  // cross-wave reduction through LDS: [wave][a][DH] for dk then dv (reuses the staging tiles)
  __syncthreads();
  float* rK = sQ;     // 4*16*128 floats = 32 KiB = exactly the sQ tile
  float* rV = sdO;
  if (a < A) {
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      *reinterpret_cast<float4*>(rK + (w * 16 + a) * DH + 16 * c + 4 * part) =
          make_float4(dk[4 * c], dk[4 * c + 1], dk[4 * c + 2], dk[4 * c + 3]);
      *reinterpret_cast<float4*>(rV + (w * 16 + a) * DH + 16 * c + 4 * part) =
          make_float4(dv[4 * c], dv[4 * c + 1], dv[4 * c + 2], dv[4 * c + 3]);
    }
  }
  __syncthreads();
  for (int idx = threadIdx.x; idx < A * DH; idx += 256) {
    const int aa = idx / DH, d = idx % DH;
    float sk = 0.f, sv = 0.f;
#pragma unroll
    for (int ww = 0; ww < 4; ++ww) {
      sk += rK[(ww * 16 + aa) * DH + d];
      sv += rV[(ww * 16 + aa) * DH + d];
    }
    const size_t oidx = ((size_t)n * A + aa) * D + h * DH + d;
    dka_part[oidx] = sk;
    dva_part[oidx] = sv;
  }
}

// sum the per-sequence adapter partials into the adapter rows of dqkv; finish the gate grads
template <typename T>
__global__ __launch_bounds__(256) void attn_bwd_reduce_k(const float* __restrict__ dka_part,
                                                         const float* __restrict__ dva_part,
                                                         const float* __restrict__ gate_part,
                                                         const float* __restrict__ gate1, T* __restrict__ dqkv,
                                                         float* __restrict__ dgate1, float* __restrict__ dgate2,
                                                         int n_seq, int S, int H, int A, int nqb) {
  const int D = H * DH;
  const size_t ld = (size_t)3 * D;
  const int total = A * D;
  for (int idx = blockIdx.x * 256 + threadIdx.x; idx < total; idx += gridDim.x * 256) {
    const int a = idx / D, col = idx % D;
    float sk = 0.f, sv = 0.f;
    for (int n = 0; n < n_seq; ++n) {
      sk += dka_part[((size_t)n * A + a) * D + col];
      sv += dva_part[((size_t)n * A + a) * D + col];
    }
    T* row = dqkv + ((size_t)n_seq * S + a) * ld;
    row[col] = from_f32<T>(0.f);
    row[D + col] = from_f32<T>(sk);
    row[2 * D + col] = from_f32<T>(sv);
  }
  if (blockIdx.x == 0 && threadIdx.x < H) {
    const int h = threadIdx.x;
    float s1 = 0.f, s2 = 0.f;
    for (int n = 0; n < n_seq; ++n)
      for (int b = 0; b < nqb; ++b) {
        const size_t p = (((size_t)n * H + h) * nqb + b) * 2;
        s1 += gate_part[p];
        s2 += gate_part[p + 1];
      }
    const float g1 = tanhf(gate1[h]);
    dgate1[h] += s1 * (1.f - g1 * g1);
    dgate2[h] += s2;
  }
}

struct BwdWs {
  size_t arrive, delta_a, delta_t, gate_part, dka, dva, total;
};
inline BwdWs bwd_ws(int n_seq, int S, int H, int A) {
  BwdWs w;
  const size_t nhs = (size_t)n_seq * H * S;
  const size_t nqb = (S + TILE - 1) / TILE;
  size_t off = 0;
  auto take = [&](size_t n) { size_t o = off; off += (n * 4 + 255) & ~(size_t)255; return o; };
  w.arrive = take(256);                 // per-head arrival counters of the fused bf16 backward: zero on first use,
                                        // left zero on return (include/fvqa.h)
  w.delta_a = take(nhs);
  w.delta_t = take(nhs);
  w.gate_part = take((size_t)n_seq * H * nqb * 2);
  w.dka = take((size_t)n_seq * A * H * DH);
  w.dva = take((size_t)n_seq * A * H * DH);
  w.total = off;
  return w;
}

inline int check_dims(int n_seq, int S, int H, int head_dim, int A, int F) {
  if (n_seq <= 0 || S <= 0 || H <= 0 || A <= 0 || F < 0) return FVQA_ESHAPE;
  if (head_dim != DH) return FVQA_ESHAPE;
  if (A > 16 || H > 256) return FVQA_ESHAPE;
  return FVQA_OK;
}

}  // namespace

// bf16 production build on the matrix cores (attn_mfma.hip); FVQA_ATTN_VALU=1 keeps the vector build
int fvqa_attn_mfma_qblocks(int S);
int fvqa_attn_fwd_mfma(const void* qkv, void* o, float* lse_a, float* lse_t, const float* gate1, const float* gate2,
                       const int32_t* vstart, const float* cos_t, const float* sin_t, int n_seq, int S, int H, int A,
                       int F, hipStream_t st);
int fvqa_attn_bwd_mfma(const void* d_o, const void* qkv, const void* o, const float* lse_a, const float* lse_t,
                       const float* gate1, const float* gate2, const int32_t* vstart, const float* cos_t,
                       const float* sin_t, void* dqkv, float* dgate1, float* dgate2, float* delta_a, float* delta_t,
                       float* gate_part, float* dka, float* dva, int* arrive, int n_seq, int S, int H, int A, int F,
                       hipStream_t st, int prerotated);
static bool use_mfma_attention() {
  static const bool v = [] { const char* e = getenv("FVQA_ATTN_VALU"); return !(e && e[0] == '1'); }();
  return v;
}

// 1 when fvqa_attn_fwd/bwd of this dtype apply RoPE themselves (cos_t/sin_t arguments): the bf16 MFMA build
extern "C" int fvqa_attn_rope_fused(int dtype) { return dtype == FVQA_H16 && use_mfma_attention() ? 1 : 0; }

extern "C" int fvqa_attn_fwd(const void* qkv, void* o, float* lse_a, float* lse_t, const float* gate1,
                             const float* gate2, const int32_t* vstart, const float* cos_t, const float* sin_t,
                             int n_seq, int seq_len, int n_heads, int head_dim, int adapter_len, int max_feats,
                             int dtype, void* stream) {
  if (!qkv || !o || !lse_a || !lse_t || !gate1 || !gate2 || !vstart) return FVQA_EINVAL;
  if (!fvqa_dtype_ok(dtype)) return FVQA_EINVAL;
  if ((cos_t == nullptr) != (sin_t == nullptr)) return FVQA_EINVAL;
  if (cos_t && !fvqa_attn_rope_fused(dtype)) return FVQA_EINVAL;      // the vector build takes rotated q,k only
  int rc = check_dims(n_seq, seq_len, n_heads, head_dim, adapter_len, max_feats);
  if (rc) return rc;
  const int nqb = (seq_len + TILE - 1) / TILE;
  dim3 grid(nqb, n_heads, n_seq), block(256);
  const size_t lds = (size_t)(2 * TILE + 2 * adapter_len) * DH * sizeof(float);
  if (dtype == FVQA_H16 && use_mfma_attention()) {
    fvqa_attn_fwd_mfma(qkv, o, lse_a, lse_t, gate1, gate2, vstart, cos_t, sin_t, n_seq, seq_len, n_heads, adapter_len,
                       max_feats, (hipStream_t)stream);
    FVQA_CHECK_LAUNCH();
    return FVQA_OK;
  }
  if (dtype == FVQA_H16)
    hipLaunchKernelGGL(attn_fwd_k<bf16_t>, grid, block, lds, (hipStream_t)stream, (const bf16_t*)qkv, (bf16_t*)o,
                       lse_a, lse_t, gate1, gate2, vstart, n_seq, seq_len, n_heads, adapter_len, max_feats);
  else
    hipLaunchKernelGGL(attn_fwd_k<float>, grid, block, lds, (hipStream_t)stream, (const float*)qkv, (float*)o, lse_a,
                       lse_t, gate1, gate2, vstart, n_seq, seq_len, n_heads, adapter_len, max_feats);
  FVQA_CHECK_LAUNCH();
  return FVQA_OK;
}

extern "C" size_t fvqa_attn_bwd_workspace(int n_seq, int seq_len, int n_heads, int head_dim, int adapter_len) {
  if (check_dims(n_seq, seq_len, n_heads, head_dim, adapter_len, 0)) return 0;
  return bwd_ws(n_seq, seq_len, n_heads, adapter_len).total;
}

static int attn_bwd_impl(const void* d_o, const void* qkv, const void* o, const float* lse_a,
                             const float* lse_t, const float* gate1, const float* gate2, const int32_t* vstart,
                             const float* cos_t, const float* sin_t, void* dqkv, float* dgate1, float* dgate2,
                             void* workspace, size_t workspace_bytes, int n_seq, int seq_len, int n_heads,
                             int head_dim, int adapter_len, int max_feats, int dtype, void* stream, int prerotated) {
  if ((cos_t == nullptr) != (sin_t == nullptr)) return FVQA_EINVAL;
  if (prerotated && !cos_t) return FVQA_EINVAL;
  if (cos_t && !fvqa_attn_rope_fused(dtype)) return FVQA_EINVAL;
  if (!d_o || !qkv || !o || !lse_a || !lse_t || !gate1 || !gate2 || !vstart || !dqkv || !dgate1 || !dgate2 ||
      !workspace)
    return FVQA_EINVAL;
  if (!fvqa_dtype_ok(dtype)) return FVQA_EINVAL;
  int rc = check_dims(n_seq, seq_len, n_heads, head_dim, adapter_len, max_feats);
  if (rc) return rc;
  const BwdWs ws = bwd_ws(n_seq, seq_len, n_heads, adapter_len);
  if (workspace_bytes < ws.total || ((uintptr_t)workspace & 15)) return FVQA_EALIGN;
  char* wb = (char*)workspace;
  float* delta_a = (float*)(wb + ws.delta_a);
  float* delta_t = (float*)(wb + ws.delta_t);
  float* gate_part = (float*)(wb + ws.gate_part);
  float* dka = (float*)(wb + ws.dka);
  float* dva = (float*)(wb + ws.dva);
  const int nqb = (seq_len + TILE - 1) / TILE;
  hipStream_t st = (hipStream_t)stream;
  dim3 block(256);
  const size_t lds_q = (size_t)(2 * TILE + 2 * adapter_len) * DH * sizeof(float);
  const size_t lds_kv = (size_t)(2 * TILE * DH + 2 * TILE) * sizeof(float);
  if (dtype == FVQA_H16 && use_mfma_attention()) {
    const int reduced = fvqa_attn_bwd_mfma(d_o, qkv, o, lse_a, lse_t, gate1, gate2, vstart, cos_t, sin_t, dqkv, dgate1,
                                           dgate2, delta_a, delta_t, gate_part, dka, dva, (int*)(wb + ws.arrive), n_seq,
                                           seq_len, n_heads, adapter_len, max_feats, st, prerotated);
    if (!reduced)
      hipLaunchKernelGGL(attn_bwd_reduce_k<bf16_t>, dim3(64), block, 0, st, dka, dva, gate_part, gate1, (bf16_t*)dqkv,
                         dgate1, dgate2, n_seq, seq_len, n_heads, adapter_len, fvqa_attn_mfma_qblocks(seq_len));
    FVQA_CHECK_LAUNCH();
    return FVQA_OK;
  }
  if (dtype == FVQA_H16) {
    typedef bf16_t T;
    hipLaunchKernelGGL(attn_bwd_dq_k<T>, dim3(nqb, n_heads, n_seq), block, lds_q, st, (const T*)d_o, (const T*)qkv,
                       (const T*)o, lse_a, lse_t, gate1, gate2, vstart, (T*)dqkv, delta_a, delta_t, gate_part, n_seq,
                       seq_len, n_heads, adapter_len, max_feats);
    hipLaunchKernelGGL(attn_bwd_dkv_k<T>, dim3(nqb + 1, n_heads, n_seq), block, lds_kv, st, (const T*)d_o,
                       (const T*)qkv, lse_a, lse_t, delta_a, delta_t, gate1, gate2, vstart, (T*)dqkv, dka, dva, n_seq,
                       seq_len, n_heads, adapter_len, max_feats, nqb);
    hipLaunchKernelGGL(attn_bwd_reduce_k<T>, dim3(64), block, 0, st, dka, dva, gate_part, gate1, (T*)dqkv, dgate1,
                       dgate2, n_seq, seq_len, n_heads, adapter_len, nqb);
  } else {
    typedef float T;
    hipLaunchKernelGGL(attn_bwd_dq_k<T>, dim3(nqb, n_heads, n_seq), block, lds_q, st, (const T*)d_o, (const T*)qkv,
                       (const T*)o, lse_a, lse_t, gate1, gate2, vstart, (T*)dqkv, delta_a, delta_t, gate_part, n_seq,
                       seq_len, n_heads, adapter_len, max_feats);
    hipLaunchKernelGGL(attn_bwd_dkv_k<T>, dim3(nqb + 1, n_heads, n_seq), block, lds_kv, st, (const T*)d_o,
                       (const T*)qkv, lse_a, lse_t, delta_a, delta_t, gate1, gate2, vstart, (T*)dqkv, dka, dva, n_seq,
                       seq_len, n_heads, adapter_len, max_feats, nqb);
    hipLaunchKernelGGL(attn_bwd_reduce_k<T>, dim3(64), block, 0, st, dka, dva, gate_part, gate1, (T*)dqkv, dgate1,
                       dgate2, n_seq, seq_len, n_heads, adapter_len, nqb);
  }
  FVQA_CHECK_LAUNCH();
  return FVQA_OK;
}

extern "C" int fvqa_attn_bwd(const void* d_o, const void* qkv, const void* o, const float* lse_a,
                             const float* lse_t, const float* gate1, const float* gate2, const int32_t* vstart,
                             const float* cos_t, const float* sin_t, void* dqkv, float* dgate1, float* dgate2,
                             void* workspace, size_t workspace_bytes, int n_seq, int seq_len, int n_heads,
                             int head_dim, int adapter_len, int max_feats, int dtype, void* stream) {
  return attn_bwd_impl(d_o, qkv, o, lse_a, lse_t, gate1, gate2, vstart, cos_t, sin_t, dqkv, dgate1, dgate2, workspace,
                       workspace_bytes, n_seq, seq_len, n_heads, head_dim, adapter_len, max_feats, dtype, stream, 0);
}

// q, k in `qkv` ALREADY ROTATED (fvqa_gemm_nt_rope); dqkv receives the gradients of the RAW projections (bf16 MFMA build)
extern "C" int fvqa_attn_bwd_rotated(const void* d_o, const void* qkv, const void* o, const float* lse_a,
                                     const float* lse_t, const float* gate1, const float* gate2, const int32_t* vstart,
                                     const float* cos_t, const float* sin_t, void* dqkv, float* dgate1, float* dgate2,
                                     void* workspace, size_t workspace_bytes, int n_seq, int seq_len, int n_heads,
                                     int head_dim, int adapter_len, int max_feats, int dtype, void* stream) {
  if (!cos_t || !sin_t || !fvqa_attn_rope_fused(dtype)) return FVQA_EINVAL;
  return attn_bwd_impl(d_o, qkv, o, lse_a, lse_t, gate1, gate2, vstart, cos_t, sin_t, dqkv, dgate1, dgate2, workspace,
                       workspace_bytes, n_seq, seq_len, n_heads, head_dim, adapter_len, max_feats, dtype, stream, 1);
}
