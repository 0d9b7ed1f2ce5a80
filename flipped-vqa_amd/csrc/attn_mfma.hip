// Adapter-gated prefix attention + causal attention on the matrix cores (bf16 production build).
//
// Same mathematics as attn.hip (reference llama/model.py:98-126); that file stays the exact-fp32
// vector-ALU build used for the parity gate. Here every product is a v_mfma_f32_16x16x32_bf16 tile
// product in the NT form C = A·Bᵀ (both operands contraction-contiguous, 16-byte fragments), fp32
// accumulation, fp32 softmax:
//   forward, per (sequence, head, 64-query block), one 16-row tile per wave, 64-key tiles:
//     S  = Q·Kᵀ           A = Q rows (registers, loaded once)      B = K rows        (LDS, row-major)
//     O += P·V            A = P (bf16, through a per-wave LDS tile) B = Vᵀ rows       (LDS, transposed
//                                                                                      while staging)
//   backward dQ kernel (query blocks):  S, dP = dO·Vᵀ (B = V rows), dQ += dS·K (B = Kᵀ rows)
//   backward dK/dV kernel (key blocks): Sᵀ = K·Qᵀ and dPᵀ = V·dOᵀ come out of the MFMA already
//     key-major, so Pᵀ and dSᵀ feed dV += Pᵀ·dO (B = dOᵀ rows) and dK += dSᵀ·Q (B = Qᵀ rows)
//     through one LDS round trip, no cross-lane transposes.
// Row statistics use the C/D layout of the 16x16 MFMA (col = lane&15, row = 4*(lane>>4)+reg): a row
// reduction is four xor-shuffles inside a 16-lane group. LDS tiles are padded (+8 bf16 per row) so
// the ds_read_b128 fragment reads of a 16-lane group fall on distinct bank slots.
// The adapter prefix (A <= 16 keys, no RoPE, own softmax scaled by tanh(gate1)) is one extra 16-key
// tile; its key/value gradients are summed over sequences by attn_bwd_reduce_k (attn.hip), exactly
// as in the vector build: no float atomics, bitwise repeatable.
#include "common.h"
#include <stdlib.h>

namespace {

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));

constexpr int DH = 128;
constexpr int QB = 64;                  // rows (queries or keys) per workgroup
constexpr int KT = 64;                  // rows per staged tile
constexpr int LDR = DH + 8;             // row-major tile leading dim (bf16 elements): 272 B
constexpr int LDT = KT + 8;             // transposed tile leading dim: 144 B
constexpr int LDP = KT + 8;             // per-wave P / dS tile leading dim
constexpr float NEG_BIG = -1e30f;

__device__ __forceinline__ f32x4 mma(const uint4& a, const uint4& b, f32x4 acc) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b),
                                                 acc, 0, 0, 0);
}
__device__ __forceinline__ float group16_max(float v) {
#pragma unroll
  for (int o = 8; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
__device__ __forceinline__ float group16_sum(float v) {
#pragma unroll
  for (int o = 8; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// stage `nrows` rows x 128 of a row-major global matrix (row stride ld elements) into sR[nrows][LDR];
// rows >= row_limit are zero-filled
__device__ __forceinline__ void stage_rows(bf16_t* sR, const bf16_t* g, size_t ld, int row0, int row_limit,
                                           int nrows) {
  for (int idx = threadIdx.x; idx < nrows * 16; idx += 256) {
    const int r = idx >> 4, c = idx & 15;
    uint4 v = make_uint4(0, 0, 0, 0);
    if (row0 + r < row_limit) v = *reinterpret_cast<const uint4*>(g + (size_t)(row0 + r) * ld + c * 8);
    *reinterpret_cast<uint4*>(sR + r * LDR + c * 8) = v;
  }
}
// same rows, stored transposed: sT[d][r] (leading dim ldt), r < nrows
__device__ __forceinline__ void stage_rows_t(bf16_t* sT, int ldt, const bf16_t* g, size_t ld, int row0,
                                             int row_limit, int nrows) {
  for (int idx = threadIdx.x; idx < nrows * 16; idx += 256) {
    const int r = idx % nrows, c = idx / nrows;     // consecutive lanes -> consecutive r: conflict-light stores
    uint4 v = make_uint4(0, 0, 0, 0);
    if (row0 + r < row_limit) v = *reinterpret_cast<const uint4*>(g + (size_t)(row0 + r) * ld + c * 8);
    const unsigned short* e = reinterpret_cast<const unsigned short*>(&v);
    unsigned short* dst = reinterpret_cast<unsigned short*>(sT);
#pragma unroll
    for (int i = 0; i < 8; ++i) dst[(c * 8 + i) * ldt + r] = e[i];
  }
}
// A/B fragment of the 16x16x32 MFMA from a row-major LDS tile: row (lane&15), 8 elements at k0+8*(lane>>4)
__device__ __forceinline__ uint4 frag(const bf16_t* s, int ld, int row0, int k0, int lane) {
  return *reinterpret_cast<const uint4*>(s + (row0 + (lane & 15)) * ld + k0 + 8 * (lane >> 4));
}
// B fragments taken COLUMN-wise from a row-major LDS tile X[k][n] with the hardware transpose read
// ds_read_b64_tr_b16: per 16-lane group it reads 4 rows x 16 columns and hands lane i column i (its 4
// rows in the 4 elements); lane 4q+p of the group supplies the address of row q, columns 4p..4p+3.
// Two reads (rows 8g..8g+3 and 8g+4..8g+7 of the 32-deep k-step, g = lane>>4) make the 8-element
// fragment B[k = 8g + j][n = n0 + (lane&15)]. Fills f[d] for the 8 column tiles n0 = 16*d; one wait.
__device__ __forceinline__ void frags_tr8(const bf16_t* s, int k0, int lane, uint4 (&f)[8]) {
  const int g = lane >> 4, i = lane & 15;
  const bf16_t* a0 = s + (k0 + 8 * g + (i >> 2)) * LDR + 4 * (i & 3);
  const unsigned addr = (unsigned)(uintptr_t)(const __attribute__((address_space(3))) bf16_t*)a0;
  uint2 lo[8], hi[8];
#define FVQA_TR(d)                                                                                       \
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(lo[d]) : "v"(addr), "i"(32 * d));            \
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(hi[d]) : "v"(addr), "i"(32 * d + 8 * LDR));
  FVQA_TR(0) FVQA_TR(1) FVQA_TR(2) FVQA_TR(3) FVQA_TR(4) FVQA_TR(5) FVQA_TR(6) FVQA_TR(7)
#undef FVQA_TR
  asm volatile("s_waitcnt lgkmcnt(0)"
               : "+v"(lo[0]), "+v"(lo[1]), "+v"(lo[2]), "+v"(lo[3]), "+v"(lo[4]), "+v"(lo[5]), "+v"(lo[6]), "+v"(lo[7]),
                 "+v"(hi[0]), "+v"(hi[1]), "+v"(hi[2]), "+v"(hi[3]), "+v"(hi[4]), "+v"(hi[5]), "+v"(hi[6]), "+v"(hi[7]));
#pragma unroll
  for (int d = 0; d < 8; ++d) f[d] = make_uint4(lo[d].x, lo[d].y, hi[d].x, hi[d].y);
}

__device__ __forceinline__ uint4 frag_g(const bf16_t* g, size_t ld, int row, int k0, int lane) {
  return *reinterpret_cast<const uint4*>(g + (size_t)row * ld + k0 + 8 * (lane >> 4));
}

// ------------------------------------------------------------------------------- forward
// TR: take the V fragments with transposing LDS reads from the row-major tile (no transposed staging)
template <bool TR>
__global__ __launch_bounds__(256) void attn_fwd_mfma_k(const bf16_t* __restrict__ qkv, bf16_t* __restrict__ o,
                                                       float* __restrict__ lse_a, float* __restrict__ lse_t,
                                                       const float* __restrict__ gate1,
                                                       const float* __restrict__ gate2,
                                                       const int32_t* __restrict__ vstart, int n_seq, int S, int H,
                                                       int A, int F) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  bf16_t* sK = reinterpret_cast<bf16_t*>(smem_raw);      // [KT][LDR]
  bf16_t* sVT = sK + KT * LDR;                            // [DH][LDT]  (TR: [KT][LDR] row-major V)
  bf16_t* sP = sVT + (TR ? KT * LDR : DH * LDT);          // [4][16][LDP]
  const int qb = blockIdx.x, h = blockIdx.y, n = blockIdx.z;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int D = H * DH;
  const size_t ld = (size_t)3 * D;
  const bf16_t* seq = qkv + (size_t)n * S * ld + h * DH;
  const int i0 = qb * QB + w * 16;                        // first query row of this wave
  const int col = lane & 15, rq = (lane >> 4) * 4;        // C/D layout: rows rq..rq+3, column col
  const float sc = rsqrtf((float)DH);

  uint4 qf[4];
  {
    int r = i0 + (lane & 15);
    r = r < S ? r : S - 1;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) qf[ks] = frag_g(seq, ld, r, 32 * ks, lane);
  }
  f32x4 oacc[8];
#pragma unroll
  for (int d = 0; d < 8; ++d) oacc[d] = f32x4{0.f, 0.f, 0.f, 0.f};
  float m[4], ls[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) { m[r] = NEG_BIG; ls[r] = 0.f; }
  const int vs = vstart[n];
  const float g2 = gate2[h];
  bf16_t* myP = sP + w * 16 * LDP;

  for (int kt = 0; kt <= qb; ++kt) {
    __syncthreads();
    stage_rows(sK, seq + D, ld, kt * KT, S, KT);
    if (TR) stage_rows(sVT, seq + 2 * D, ld, kt * KT, S, KT);
    else stage_rows_t(sVT, LDT, seq + 2 * D, ld, kt * KT, S, KT);
    __syncthreads();
    f32x4 s[4];
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) {
      s[ct] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) s[ct] = mma(qf[ks], frag(sK, LDR, 16 * ct, 32 * ks, lane), s[ct]);
    }
    float alpha[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int i = i0 + rq + r;
      float mx = NEG_BIG;
#pragma unroll
      for (int ct = 0; ct < 4; ++ct) {
        const int j = kt * KT + 16 * ct + col;
        float v = s[ct][r] * sc;
        if (vs >= 0 && i >= vs + F && j >= vs && j < vs + F) v += g2;
        v = (j <= i && j < S) ? v : NEG_BIG;
        s[ct][r] = v;
        mx = fmaxf(mx, v);
      }
      mx = group16_max(mx);
      const float mn = fmaxf(m[r], mx);
      alpha[r] = __expf(m[r] - mn);
      m[r] = mn;
      float rs = 0.f;
#pragma unroll
      for (int ct = 0; ct < 4; ++ct) {
        const float p = (s[ct][r] > 0.5f * NEG_BIG) ? __expf(s[ct][r] - mn) : 0.f;
        rs += p;
        myP[(rq + r) * LDP + 16 * ct + col] = __float2bfloat16(p);
      }
      ls[r] = ls[r] * alpha[r] + rs;              // per-lane partial row sum (reduced at the end)
    }
#pragma unroll
    for (int d = 0; d < 8; ++d)
#pragma unroll
      for (int r = 0; r < 4; ++r) oacc[d][r] *= alpha[r];
    __syncthreads();                               // P tile visible (own wave), nobody still reads it
#pragma unroll
    for (int k2 = 0; k2 < 2; ++k2) {
      const uint4 pf = frag(myP, LDP, 0, 32 * k2, lane);
      if (TR) {
        uint4 vf8[8];
        frags_tr8(sVT, 32 * k2, lane, vf8);
#pragma unroll
        for (int d = 0; d < 8; ++d) oacc[d] = mma(pf, vf8[d], oacc[d]);
      } else {
#pragma unroll
        for (int d = 0; d < 8; ++d) oacc[d] = mma(pf, frag(sVT, LDT, 16 * d, 32 * k2, lane), oacc[d]);
      }
    }
  }
  float lt[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const float l = group16_sum(ls[r]);
    const float inv = 1.f / l;
    lt[r] = m[r] + __logf(l);
#pragma unroll
    for (int d = 0; d < 8; ++d) oacc[d][r] *= inv;
  }

  // ---- adapter prefix tile (keys padded to 16 for S, to 32 for the P·V k-step)
  const bf16_t* arow = qkv + (size_t)n_seq * S * ld + h * DH;
  __syncthreads();
  stage_rows(sK, arow + D, ld, 0, A, 16);
  if (TR) stage_rows(sVT, arow + 2 * D, ld, 0, A, 32);
  else stage_rows_t(sVT, LDT, arow + 2 * D, ld, 0, A, 32);
  __syncthreads();
  f32x4 sa = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) sa = mma(qf[ks], frag(sK, LDR, 0, 32 * ks, lane), sa);
  const float g1 = tanhf(gate1[h]);
  float la[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const float v = (col < A) ? sa[r] * sc : NEG_BIG;
    const float mx = group16_max(v);
    const float e = (col < A) ? __expf(v - mx) : 0.f;
    const float sum = group16_sum(e);
    la[r] = mx + __logf(sum);
    myP[(rq + r) * LDP + col] = __float2bfloat16(g1 * e / sum);
    myP[(rq + r) * LDP + 16 + col] = __float2bfloat16(0.f);
  }
  __syncthreads();
  {
    const uint4 pf = frag(myP, LDP, 0, 0, lane);
    if (TR) {
      uint4 vf8[8];
      frags_tr8(sVT, 0, lane, vf8);
#pragma unroll
      for (int d = 0; d < 8; ++d) oacc[d] = mma(pf, vf8[d], oacc[d]);
    } else {
#pragma unroll
      for (int d = 0; d < 8; ++d) oacc[d] = mma(pf, frag(sVT, LDT, 16 * d, 0, lane), oacc[d]);
    }
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int i = i0 + rq + r;
    if (i < S) {
      bf16_t* orow = o + ((size_t)n * S + i) * D + h * DH;
#pragma unroll
      for (int d = 0; d < 8; ++d) orow[16 * d + col] = __float2bfloat16(oacc[d][r]);
      if (col == 0) {
        lse_a[((size_t)n * H + h) * S + i] = la[r];
        lse_t[((size_t)n * H + h) * S + i] = lt[r];
      }
    }
  }
}

// ------------------------------------------------------------------------------- backward: dQ
__global__ __launch_bounds__(256) void attn_bwd_dq_mfma_k(
    const bf16_t* __restrict__ d_o, const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ o,
    const float* __restrict__ lse_a, const float* __restrict__ lse_t, const float* __restrict__ gate1,
    const float* __restrict__ gate2, const int32_t* __restrict__ vstart, bf16_t* __restrict__ dqkv,
    float* __restrict__ delta_a, float* __restrict__ delta_t, float* __restrict__ gate_part, int n_seq, int S, int H,
    int A, int F) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  bf16_t* sK = reinterpret_cast<bf16_t*>(smem_raw);      // [KT][LDR]   K rows (also read column-wise for dS·K)
  bf16_t* sV = sK + KT * LDR;                             // [KT][LDR]   V rows
  bf16_t* sP = sV + KT * LDR;                             // [4][16][LDP] dS tile per wave
  __shared__ float red[8];
  const int qb = blockIdx.x, h = blockIdx.y, n = blockIdx.z;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int D = H * DH;
  const size_t ld = (size_t)3 * D;
  const bf16_t* seq = qkv + (size_t)n * S * ld + h * DH;
  const int i0 = qb * QB + w * 16;
  const int col = lane & 15, rq = (lane >> 4) * 4;
  const float sc = rsqrtf((float)DH);
  const size_t sbase = ((size_t)n * H + h) * S;

  uint4 qf[4], dof[4];
  {
    int r = i0 + (lane & 15);
    r = r < S ? r : S - 1;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      qf[ks] = frag_g(seq, ld, r, 32 * ks, lane);
      dof[ks] = frag_g(d_o + (size_t)n * S * D + h * DH, (size_t)D, r, 32 * ks, lane);
    }
  }
  // row statistics in C/D layout: rows i0+rq+r
  float lsa[4], lst[4], dtot[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    int i = i0 + rq + r;
    i = i < S ? i : S - 1;
    lsa[r] = lse_a[sbase + i];
    lst[r] = lse_t[sbase + i];
    // dO·O over the row: each lane of the 16-lane group takes 8 of the 128 dims
    const bf16_t* dor = d_o + ((size_t)n * S + i) * D + h * DH + col * 8;
    const bf16_t* orr = o + ((size_t)n * S + i) * D + h * DH + col * 8;
    float a[8], b[8];
    {
      const uint4 t = *reinterpret_cast<const uint4*>(dor), u = *reinterpret_cast<const uint4*>(orr);
      const unsigned tw[4] = {t.x, t.y, t.z, t.w}, uw[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        a[2 * k] = __uint_as_float(tw[k] << 16); a[2 * k + 1] = __uint_as_float(tw[k] & 0xFFFF0000u);
        b[2 * k] = __uint_as_float(uw[k] << 16); b[2 * k + 1] = __uint_as_float(uw[k] & 0xFFFF0000u);
      }
    }
    float acc = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) acc += a[k] * b[k];
    dtot[r] = group16_sum(acc);
  }
  const float g1 = tanhf(gate1[h]);
  const float g2 = gate2[h];
  const int vs = vstart[n];
  bf16_t* myP = sP + w * 16 * LDP;
  f32x4 dq[8];
#pragma unroll
  for (int d = 0; d < 8; ++d) dq[d] = f32x4{0.f, 0.f, 0.f, 0.f};

  // ---- adapter tile: dS_a, delta_a, d tanh-gate partial
  const bf16_t* arow = qkv + (size_t)n_seq * S * ld + h * DH;
  stage_rows(sK, arow + D, ld, 0, A, 32);                 // rows >= A are zero (32 = one MFMA k-step)
  stage_rows(sV, arow + 2 * D, ld, 0, A, 16);
  __syncthreads();
  float da[4], dg1 = 0.f;
  {
    f32x4 sa = f32x4{0.f, 0.f, 0.f, 0.f}, dpa = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      sa = mma(qf[ks], frag(sK, LDR, 0, 32 * ks, lane), sa);
      dpa = mma(dof[ks], frag(sV, LDR, 0, 32 * ks, lane), dpa);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float p = (col < A) ? __expf(sa[r] * sc - lsa[r]) : 0.f;
      const float dov = (col < A) ? dpa[r] : 0.f;
      const bool live = i0 + rq + r < S;
      if (live) dg1 += dov * p;
      da[r] = group16_sum(p * g1 * dov);
      const float ds = p * (g1 * dov - da[r]);
      myP[(rq + r) * LDP + col] = __float2bfloat16(ds);
      myP[(rq + r) * LDP + 16 + col] = __float2bfloat16(0.f);
    }
  }
  __syncthreads();
  {
    const uint4 pf = frag(myP, LDP, 0, 0, lane);
    uint4 kf8[8];
    frags_tr8(sK, 0, lane, kf8);
#pragma unroll
    for (int d = 0; d < 8; ++d) dq[d] = mma(pf, kf8[d], dq[d]);
  }
  float dt[4], dg2 = 0.f;
#pragma unroll
  for (int r = 0; r < 4; ++r) dt[r] = dtot[r] - da[r];

  for (int kt = 0; kt <= qb; ++kt) {
    __syncthreads();
    stage_rows(sK, seq + D, ld, kt * KT, S, KT);
    stage_rows(sV, seq + 2 * D, ld, kt * KT, S, KT);
    __syncthreads();
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) {
      f32x4 s = f32x4{0.f, 0.f, 0.f, 0.f}, dp = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        s = mma(qf[ks], frag(sK, LDR, 16 * ct, 32 * ks, lane), s);
        dp = mma(dof[ks], frag(sV, LDR, 16 * ct, 32 * ks, lane), dp);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int i = i0 + rq + r;
        const int j = kt * KT + 16 * ct + col;
        float v = s[r] * sc;
        const bool inwin = vs >= 0 && i >= vs + F && j >= vs && j < vs + F;
        if (inwin) v += g2;
        const float p = (j <= i && j < S && i < S) ? __expf(v - lst[r]) : 0.f;
        const float ds = p * (dp[r] - dt[r]);
        if (inwin) dg2 += ds;
        myP[(rq + r) * LDP + 16 * ct + col] = __float2bfloat16(ds);
      }
    }
    __syncthreads();
#pragma unroll
    for (int k2 = 0; k2 < 2; ++k2) {
      const uint4 pf = frag(myP, LDP, 0, 32 * k2, lane);
      uint4 kf8[8];
      frags_tr8(sK, 32 * k2, lane, kf8);
#pragma unroll
      for (int d = 0; d < 8; ++d) dq[d] = mma(pf, kf8[d], dq[d]);
    }
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int i = i0 + rq + r;
    if (i < S) {
      bf16_t* row = dqkv + ((size_t)n * S + i) * ld + h * DH;
#pragma unroll
      for (int d = 0; d < 8; ++d) row[16 * d + col] = __float2bfloat16(dq[d][r] * sc);
      if (col == 0) {
        delta_a[sbase + i] = da[r];
        delta_t[sbase + i] = dt[r];
      }
    }
  }
  const float b1 = block_sum_256(dg1, red);
  const float b2 = block_sum_256(dg2, red + 4);
  if (threadIdx.x == 0) {
    const size_t pidx = (((size_t)n * H + h) * gridDim.x + qb) * 2;
    gate_part[pidx] = b1;
    gate_part[pidx + 1] = b2;
  }
}

// ------------------------------------------------------------------------------- backward: dK, dV
// blockIdx.x < nkb: 64 text keys (one 16-key tile per wave); blockIdx.x == nkb: the adapter keys
// (16-key tile, every wave takes the query tiles t ≡ w mod 4 and the partial sums meet in LDS).
__global__ __launch_bounds__(256) void attn_bwd_dkv_mfma_k(
    const bf16_t* __restrict__ d_o, const bf16_t* __restrict__ qkv, const float* __restrict__ lse_a,
    const float* __restrict__ lse_t, const float* __restrict__ delta_a, const float* __restrict__ delta_t,
    const float* __restrict__ gate1, const float* __restrict__ gate2, const int32_t* __restrict__ vstart,
    bf16_t* __restrict__ dqkv, float* __restrict__ dka_part, float* __restrict__ dva_part, int n_seq, int S, int H,
    int A, int F, int nkb) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  bf16_t* sQ = reinterpret_cast<bf16_t*>(smem_raw);      // [KT][LDR]  Q rows  (row reads for S^T, column reads for dK)
  bf16_t* sdO = sQ + KT * LDR;                            // [KT][LDR]  dO rows (row reads for dP^T, column reads for dV)
  bf16_t* sP = sdO + KT * LDR;                            // [4][2][16][LDP]  P^T and dS^T per wave
  float* sL = reinterpret_cast<float*>(sP + 4 * 2 * 16 * LDP);   // [KT] lse, [KT] delta
  float* sDl = sL + KT;
  const int kb = blockIdx.x, h = blockIdx.y, n = blockIdx.z;
  const bool adapter = kb == nkb;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int D = H * DH;
  const size_t ld = (size_t)3 * D;
  const bf16_t* seq = qkv + (size_t)n * S * ld + h * DH;
  const bf16_t* dob = d_o + (size_t)n * S * D + h * DH;
  const int col = lane & 15, rq = (lane >> 4) * 4;        // C/D layout: key rows rq..rq+3, query column col
  const float sc = rsqrtf((float)DH);
  const size_t sbase = ((size_t)n * H + h) * S;
  bf16_t* myPT = sP + w * 2 * 16 * LDP;
  bf16_t* myST = myPT + 16 * LDP;
  f32x4 dk[8], dv[8];
#pragma unroll
  for (int d = 0; d < 8; ++d) { dk[d] = f32x4{0.f, 0.f, 0.f, 0.f}; dv[d] = f32x4{0.f, 0.f, 0.f, 0.f}; }
  const int nqt = (S + KT - 1) / KT;
  const float g1 = tanhf(gate1[h]);

  // fragments of this wave's 16 keys (A operands of S^T and dP^T)
  uint4 kf[4], vf[4];
  const int j0 = adapter ? 0 : kb * QB + w * 16;
  {
    const bf16_t* kbase = adapter ? qkv + (size_t)n_seq * S * ld + h * DH : seq;
    const int limit = adapter ? A : S;
    int r = j0 + (lane & 15);
    const bool ok = r < limit;
    r = ok ? r : limit - 1;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      kf[ks] = frag_g(kbase + D, ld, r, 32 * ks, lane);
      vf[ks] = frag_g(kbase + 2 * D, ld, r, 32 * ks, lane);
      if (!ok) { kf[ks] = make_uint4(0, 0, 0, 0); vf[ks] = make_uint4(0, 0, 0, 0); }
    }
  }
  const int vs = vstart[n];
  const float g2 = gate2[h];
  const int t_first = adapter ? 0 : kb;
  for (int t = t_first; t < nqt; ++t) {
    __syncthreads();
    stage_rows(sQ, seq, ld, t * KT, S, KT);
    stage_rows(sdO, dob, (size_t)D, t * KT, S, KT);
    if (threadIdx.x < KT) {
      const int ii = min(t * KT + (int)threadIdx.x, S - 1);
      sL[threadIdx.x] = adapter ? lse_a[sbase + ii] : lse_t[sbase + ii];
      sDl[threadIdx.x] = adapter ? delta_a[sbase + ii] : delta_t[sbase + ii];
    }
    __syncthreads();
    const bool mine = !adapter || (t & 3) == w;      // adapter block: query tiles are dealt to the waves
    if (mine) {
#pragma unroll
      for (int ct = 0; ct < 4; ++ct) {                // 16-query column tiles of this 64-query tile
        f32x4 s = f32x4{0.f, 0.f, 0.f, 0.f}, dp = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
          s = mma(kf[ks], frag(sQ, LDR, 16 * ct, 32 * ks, lane), s);
          dp = mma(vf[ks], frag(sdO, LDR, 16 * ct, 32 * ks, lane), dp);
        }
        const int i = t * KT + 16 * ct + col;          // query of this lane's column
        const float lse = sL[16 * ct + col], dl = sDl[16 * ct + col];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int j = j0 + rq + r;                   // key row
          float v = s[r] * sc;
          float p, ds;
          if (adapter) {
            p = (j < A && i < S) ? __expf(v - lse) : 0.f;
            ds = p * (g1 * dp[r] - dl);
            p *= g1;
          } else {
            if (vs >= 0 && i >= vs + F && j >= vs && j < vs + F) v += g2;
            p = (j <= i && i < S && j < S) ? __expf(v - lse) : 0.f;
            ds = p * (dp[r] - dl);
          }
          myPT[(rq + r) * LDP + 16 * ct + col] = __float2bfloat16(p);
          myST[(rq + r) * LDP + 16 * ct + col] = __float2bfloat16(ds);
        }
      }
    }
    __syncthreads();
    if (mine) {
#pragma unroll
      for (int k2 = 0; k2 < 2; ++k2) {
        const uint4 pf = frag(myPT, LDP, 0, 32 * k2, lane);
        const uint4 sf = frag(myST, LDP, 0, 32 * k2, lane);
        uint4 t8[8];
        frags_tr8(sdO, 32 * k2, lane, t8);
#pragma unroll
        for (int d = 0; d < 8; ++d) dv[d] = mma(pf, t8[d], dv[d]);
        frags_tr8(sQ, 32 * k2, lane, t8);
#pragma unroll
        for (int d = 0; d < 8; ++d) dk[d] = mma(sf, t8[d], dk[d]);
      }
    }
  }
  if (!adapter) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int j = j0 + rq + r;
      if (j < S) {
        bf16_t* row = dqkv + ((size_t)n * S + j) * ld + h * DH;
#pragma unroll
        for (int d = 0; d < 8; ++d) {
          row[D + 16 * d + col] = __float2bfloat16(dk[d][r] * sc);
          row[2 * D + 16 * d + col] = __float2bfloat16(dv[d][r]);
        }
      }
    }
    return;
  }
  // adapter block: sum the four waves' partial tiles through LDS ([wave][16][128] fp32 x 2 = 64 KiB,
  // the Q/dO staging area is free now), then one fp32 partial per sequence for the batch reduction
  __syncthreads();
  float* rK = reinterpret_cast<float*>(smem_raw);
  float* rV = rK + 4 * 16 * DH;
#pragma unroll
  for (int r = 0; r < 4; ++r)
#pragma unroll
    for (int d = 0; d < 8; ++d) {
      rK[(w * 16 + rq + r) * DH + 16 * d + col] = dk[d][r] * sc;
      rV[(w * 16 + rq + r) * DH + 16 * d + col] = dv[d][r];
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

constexpr size_t FWD_LDS = (size_t)(KT * LDR + DH * LDT + 4 * 16 * LDP) * 2;
constexpr size_t FWD_LDS_TR = (size_t)(2 * KT * LDR + 4 * 16 * LDP) * 2;
constexpr size_t DQ_LDS = (size_t)(2 * KT * LDR + 4 * 16 * LDP) * 2;
constexpr size_t DKV_LDS_A = (size_t)(2 * KT * LDR + 4 * 2 * 16 * LDP) * 2 + 2 * KT * 4;
constexpr size_t DKV_LDS_B = (size_t)2 * 4 * 16 * DH * 4;
constexpr size_t DKV_LDS = DKV_LDS_A > DKV_LDS_B ? DKV_LDS_A : DKV_LDS_B;

}  // namespace

// launched from attn.hip's C entry points when dtype == bf16 (workspace layout shared with the vector build)
int fvqa_attn_fwd_mfma(const void* qkv, void* o, float* lse_a, float* lse_t, const float* gate1, const float* gate2,
                       const int32_t* vstart, int n_seq, int S, int H, int A, int F, hipStream_t st) {
  const int nqb = (S + QB - 1) / QB;
  static const bool tr = [] { const char* e = getenv("FVQA_ATTN_TR"); return !(e && e[0] == '0'); }();
  if (tr) {
    hipLaunchKernelGGL(attn_fwd_mfma_k<true>, dim3(nqb, H, n_seq), dim3(256), FWD_LDS_TR, st, (const bf16_t*)qkv,
                       (bf16_t*)o, lse_a, lse_t, gate1, gate2, vstart, n_seq, S, H, A, F);
  } else {
    hipLaunchKernelGGL(attn_fwd_mfma_k<false>, dim3(nqb, H, n_seq), dim3(256), FWD_LDS, st, (const bf16_t*)qkv,
                       (bf16_t*)o, lse_a, lse_t, gate1, gate2, vstart, n_seq, S, H, A, F);
  }
  return 0;
}

int fvqa_attn_bwd_mfma(const void* d_o, const void* qkv, const void* o, const float* lse_a, const float* lse_t,
                       const float* gate1, const float* gate2, const int32_t* vstart, void* dqkv, float* delta_a,
                       float* delta_t, float* gate_part, float* dka, float* dva, int n_seq, int S, int H, int A, int F,
                       hipStream_t st) {
  const int nqb = (S + QB - 1) / QB;
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute((const void*)attn_bwd_dq_mfma_k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)DQ_LDS);
    (void)hipFuncSetAttribute((const void*)attn_bwd_dkv_mfma_k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)DKV_LDS);
    attr = true;
  }
  hipLaunchKernelGGL(attn_bwd_dq_mfma_k, dim3(nqb, H, n_seq), dim3(256), DQ_LDS, st, (const bf16_t*)d_o,
                     (const bf16_t*)qkv, (const bf16_t*)o, lse_a, lse_t, gate1, gate2, vstart, (bf16_t*)dqkv, delta_a,
                     delta_t, gate_part, n_seq, S, H, A, F);
  hipLaunchKernelGGL(attn_bwd_dkv_mfma_k, dim3(nqb + 1, H, n_seq), dim3(256), DKV_LDS, st, (const bf16_t*)d_o,
                     (const bf16_t*)qkv, lse_a, lse_t, delta_a, delta_t, gate1, gate2, vstart, (bf16_t*)dqkv, dka, dva,
                     n_seq, S, H, A, F, nqb);
  return 0;
}
