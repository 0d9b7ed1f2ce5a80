// Adapter-gated prefix attention + causal attention on the matrix cores (bf16 production build).
//
// Same mathematics as attn.hip (reference llama/model.py:98-126); that file stays the exact-fp32
// vector-ALU build used for the parity gate. Here every product is a v_mfma_f32_16x16x32_bf16 tile
// product in the NT form D = X·Yᵀ (D rows from X, D columns from Y; both operands contraction-
// contiguous 16-byte fragments), fp32 accumulation, fp32 softmax.
//
// One workgroup = 8 waves = 128 queries (forward, dQ) or 128 keys (dK/dV) of one (sequence, head);
// a wave owns 16 of them; the other side is staged in 128-row tiles (one tile covers S = 128) and
// consumed in groups of 32 rows = one MFMA k-step.
//
// No LDS round trip for P / dS: the score blocks are produced TRANSPOSED with respect to the product
// that consumes them, so the C/D registers of two 16-row blocks (lane: column i = lane&15, rows
// 4g..4g+3 of each block, g = lane>>4) ARE the 8-element operand fragment of the next MFMA, with the
// contraction index running in the order (4g+r | 16+4g+r). The matching operand is fetched from the
// row-major LDS tile with ds_read_b64_tr_b16 at exactly those rows (frags_tr_perm):
//   forward  (wave = 16 queries):  Sᵀ = K·Qᵀ  ->  Oᵀ += Vᵀ·P          (P from registers)
//   dQ       (wave = 16 queries):  Sᵀ, dPᵀ = V·dOᵀ  ->  dQᵀ += Kᵀ·dS   (dS from registers)
//   dK/dV    (wave = 16 keys):     S = Q·Kᵀ, dP = dO·Vᵀ -> dVᵀ += dOᵀ·P, dKᵀ += Qᵀ·dS
// Every output therefore lands as 4 consecutive head dims per lane (8-byte stores), row statistics
// are per-lane scalars (two xor-shuffles across g), and RoPE fuses for free: the rotation pairs
// (2i, 2i+1) sit in one lane both in the 16-byte loads (q, k rotated while loading / staging; values
// identical to the separate RoPE pass: bf16 in, fp32 rotate, bf16 out) and in the dQ / dK stores
// (conjugate rotation on the fp32 accumulators). cos_t == NULL means q,k arrive already rotated.
// The adapter prefix (A <= 16 keys, no RoPE, own softmax scaled by tanh(gate1)) is one extra 16-key
// block; its key/value gradients are summed over sequences by attn_bwd_reduce_k (attn.hip), exactly
// as in the vector build: no float atomics, bitwise repeatable.
#include "common.h"
#include <stdlib.h>

namespace {

constexpr int DH = 128;
constexpr int BQ = 128;                 // queries / keys per workgroup = rows per staged tile
// Row pitch of the LDS tiles (bf16 elements). 288 bytes = 72 dwords: consecutive rows start 8 banks apart, so the 8 rows x 32
// bytes that one 32-lane half of a ds_read_b64_tr_b16 takes (rows rb+4g+q, g in {0,1}, q < 4: 8 dwords each) cover 64 distinct
// banks, and the ds_read_b128 row fragments stay conflict-free as well. The 272-byte pitch of rounds 1-3 (68 dwords: rows 4
// banks apart) made those transposed reads 2-way: SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE 0.52 in the fused backward, 0.38
// in the forward (profiles/r03_pmc_mfma_lds.json). -DFVQA_ATTN_LDR=136 rebuilds the old pitch for A/B runs.
#ifndef FVQA_ATTN_LDR
#define FVQA_ATTN_LDR 144
#endif
constexpr int LDR = FVQA_ATTN_LDR;      // row-major tile leading dim (bf16 elements)
// fp32 partial blocks of the adapter keys' dK / dV ([K range or wave][16 rows][DH]): rows DH + 4 floats apart. With DH floats
// (512 bytes) every row of a 16-row fragment store started on the same bank — 16-way conflicts on the f32x4 stores, which were the
// whole 0.28 LDS conflict fraction of the fused backward in rounds 3-4 (r04_pmc_mfma_lds.json) — with 528 bytes the 16 rows of a
// store spread over all 64 banks.
#ifdef FVQA_ATTN_ADP_PITCH_OLD            // (A/B builds only: the rounds 1-4 pitch)
constexpr int ADP_PITCH = DH;
#else
constexpr int ADP_PITCH = DH + 4;
#endif
constexpr int HP = DH / 2;              // rotation pairs per head
constexpr float NEG_BIG = -1e30f;

// Tuning builds (-DFVQA_ATTN_STAMPS): 100 MHz timestamps of each workgroup's phases in the fused backward kernel
// (tools/attn_stamps.py); the shipping build compiles none of it.
#ifdef FVQA_ATTN_STAMPS
__device__ unsigned long long g_attn_stamps[1024 * 16];
#define AT_STAMP(slot) do { if (threadIdx.x == 0) g_attn_stamps[(size_t)((blockIdx.y * gridDim.x + blockIdx.x) & 1023) * 16 + (slot)] = __builtin_amdgcn_s_memrealtime(); } while (0)
// forward kernel: per workgroup, thread 0 (wave 0) and thread 448 (wave 7) sum the time they spend in each phase of the key-tile
// loop (slots 0-7 / 8-15: start, commit, barrier after commit, key groups, barrier before commit, end, tiles walked, query block)
#define AT_NOW() __builtin_amdgcn_s_memrealtime()
#define AT_FWD_DECL unsigned long long at_t = 0, at_commit = 0, at_bar2 = 0, at_groups = 0, at_bar1 = 0, at_start = 0; \
  const bool at_me = threadIdx.x == 0 || threadIdx.x == 448; if (at_me) { at_start = AT_NOW(); at_t = at_start; }
#define AT_FWD_ADD(acc) do { if (at_me) { const unsigned long long n_ = AT_NOW(); acc += n_ - at_t; at_t = n_; } } while (0)
#define AT_FWD_DUMP(tiles, qb_) do { if (at_me) { unsigned long long* p_ = g_attn_stamps + \
    (size_t)(((blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) & 1023) * 16 + (threadIdx.x == 0 ? 0 : 8); \
    p_[0] = at_start; p_[1] = at_commit; p_[2] = at_bar2; p_[3] = at_groups; p_[4] = at_bar1; p_[5] = AT_NOW(); \
    p_[6] = (tiles); p_[7] = (qb_); } } while (0)
#else
#define AT_STAMP(slot) do { } while (0)
#define AT_FWD_DECL
#define AT_FWD_ADD(acc) do { } while (0)
#define AT_FWD_DUMP(tiles, qb_) do { } while (0)
#endif

// D[4g+r][lane&15] += sum_k X[4g+r][k] * Y[lane&15][k]
__device__ __forceinline__ f32x4 mma(const uint4& x, const uint4& y, f32x4 acc) {
  return FVQA_MFMA_H16_16x16x32(__builtin_bit_cast(h16x8_t, x), __builtin_bit_cast(h16x8_t, y),
                                                 acc, 0, 0, 0);
}
__device__ __forceinline__ float across_g_max(float v) {      // over the 4 lanes that share lane&15
  v = fmaxf(v, __shfl_xor(v, 16, 64));
  return fmaxf(v, __shfl_xor(v, 32, 64));
}
__device__ __forceinline__ float across_g_sum(float v) {
  v += __shfl_xor(v, 16, 64);
  return v + __shfl_xor(v, 32, 64);
}
__device__ __forceinline__ unsigned pack2(float a, float b) {
  return (unsigned)f32_to_bf16_bits(a) | ((unsigned)f32_to_bf16_bits(b) << 16);
}
// two C/D blocks (rows 4g+r of row-block 0 and of row-block 1) -> the 8-element operand fragment
__device__ __forceinline__ uint4 pack_blocks(const float (&a)[4], const float (&b)[4]) {
  return make_uint4(pack2(a[0], a[1]), pack2(a[2], a[3]), pack2(b[0], b[1]), pack2(b[2], b[3]));
}
// RoPE of 8 consecutive head dims (4 pairs) of one row: bf16 in, fp32 rotate, bf16 out (the arithmetic of
// rope_qk_k in rowops.hip); c4/s4 point at the row's table entries of the first pair. sign = -1: conjugate.
__device__ __forceinline__ uint4 rope8(const uint4& v, const float* c4, const float* s4) {
  const float4 c = *reinterpret_cast<const float4*>(c4), s = *reinterpret_cast<const float4*>(s4);
  const unsigned w[4] = {v.x, v.y, v.z, v.w};
  const float cc[4] = {c.x, c.y, c.z, c.w}, ss[4] = {s.x, s.y, s.z, s.w};
  unsigned o[4];
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    const float e = h16_lo(w[p]), d = h16_hi(w[p]);
    o[p] = pack2(e * cc[p] - d * ss[p], e * ss[p] + d * cc[p]);
  }
  return make_uint4(o[0], o[1], o[2], o[3]);
}

// Staging of NR rows x 128 of a row-major global matrix (row stride ld elements) into sR[NR][LDR] in two
// halves, so that a kernel can put ALL its global loads in flight before the first wait: tile_load issues the
// 16-byte loads into registers (rows >= row_limit are zero), tile_commit rotates (ROPE: tables of position
// = row index) and writes LDS.
template <int NR> struct TileRegs { uint4 v[(NR * 16 + 511) / 512]; };
template <int NR>
__device__ __forceinline__ void tile_load(TileRegs<NR>& t, const bf16_t* g, size_t ld, int row0, int row_limit) {
#pragma unroll
  for (int u = 0; u < (NR * 16 + 511) / 512; ++u) {
    const int idx = threadIdx.x + u * 512;
    const int r = idx >> 4, c = idx & 15;
    t.v[u] = make_uint4(0, 0, 0, 0);
    if (idx < NR * 16 && row0 + r < row_limit)
      t.v[u] = *reinterpret_cast<const uint4*>(g + (size_t)(row0 + r) * ld + c * 8);
  }
}
template <bool ROPE, int NR>
__device__ __forceinline__ void tile_commit(TileRegs<NR>& t, bf16_t* sR, int row0, int row_limit, const float* cs,
                                            const float* sn) {
#pragma unroll
  for (int u = 0; u < (NR * 16 + 511) / 512; ++u) {
    const int idx = threadIdx.x + u * 512;
    const int r = idx >> 4, c = idx & 15;
    if (idx < NR * 16) {
      if (ROPE && row0 + r < row_limit)
        t.v[u] = rope8(t.v[u], cs + (size_t)(row0 + r) * HP + 4 * c, sn + (size_t)(row0 + r) * HP + 4 * c);
      *reinterpret_cast<uint4*>(sR + r * LDR + c * 8) = t.v[u];
    }
  }
}
// operand fragment from a row-major LDS tile: row (lane&15) of the 16-row block at row0, 8 elements at k0+8g
__device__ __forceinline__ uint4 frag(const bf16_t* s, int row0, int k0, int lane) {
  return *reinterpret_cast<const uint4*>(s + (row0 + (lane & 15)) * LDR + k0 + 8 * (lane >> 4));
}
__device__ __forceinline__ uint4 frag_g(const bf16_t* g, size_t ld, int row, int k0, int lane) {
  return *reinterpret_cast<const uint4*>(g + (size_t)row * ld + k0 + 8 * (lane >> 4));
}
// Operand fragments taken COLUMN-wise from a row-major LDS tile X[row][d] with the hardware transpose read
// ds_read_b64_tr_b16 (per 16-lane group: 4 rows x 16 columns in, lane i gets column i, its 4 rows in the 4
// elements; lane 4q+p of the group supplies the address of row q, columns 4p..4p+3). For each of the 8
// column blocks dt: f[dt] = { X[rb+4g+e][16dt+i] (e=0..3), X[rb+16+4g+e][16dt+i] (e=0..3) } — the k order of
// pack_blocks. HI = false leaves the second half zero (16-row adapter block).
struct TrRegs { uint2 lo[8], hi[8]; };
// issue only (inline asm, invisible to the compiler's waitcnt pass): the reads return while other work runs
template <bool HI>
__device__ __forceinline__ void tr_issue(const bf16_t* s, int rb, int lane, TrRegs& t) {
  const int g = lane >> 4, i = lane & 15;
  const bf16_t* a0 = s + (rb + 4 * g + (i >> 2)) * LDR + 4 * (i & 3);
  const unsigned addr = (unsigned)(uintptr_t)(const __attribute__((address_space(3))) bf16_t*)a0;
#define FVQA_TR(d)                                                                                        \
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(t.lo[d]) : "v"(addr), "i"(32 * d));           \
  if (HI) asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(t.hi[d]) : "v"(addr), "i"(32 * d + 32 * LDR));
  FVQA_TR(0) FVQA_TR(1) FVQA_TR(2) FVQA_TR(3) FVQA_TR(4) FVQA_TR(5) FVQA_TR(6) FVQA_TR(7)
#undef FVQA_TR
}
// wait for everything issued so far and assemble the 8 fragments
template <bool HI>
__device__ __forceinline__ void tr_collect(TrRegs& t, uint4 (&f)[8]) {
  if (HI) {
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+v"(t.lo[0]), "+v"(t.lo[1]), "+v"(t.lo[2]), "+v"(t.lo[3]), "+v"(t.lo[4]), "+v"(t.lo[5]),
                   "+v"(t.lo[6]), "+v"(t.lo[7]), "+v"(t.hi[0]), "+v"(t.hi[1]), "+v"(t.hi[2]), "+v"(t.hi[3]),
                   "+v"(t.hi[4]), "+v"(t.hi[5]), "+v"(t.hi[6]), "+v"(t.hi[7]));
#pragma unroll
    for (int d = 0; d < 8; ++d) f[d] = make_uint4(t.lo[d].x, t.lo[d].y, t.hi[d].x, t.hi[d].y);
  } else {
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+v"(t.lo[0]), "+v"(t.lo[1]), "+v"(t.lo[2]), "+v"(t.lo[3]), "+v"(t.lo[4]), "+v"(t.lo[5]),
                   "+v"(t.lo[6]), "+v"(t.lo[7]));
#pragma unroll
    for (int d = 0; d < 8; ++d) f[d] = make_uint4(t.lo[d].x, t.lo[d].y, 0u, 0u);
  }
}
template <bool HI>
__device__ __forceinline__ void frags_tr_perm(const bf16_t* s, int rb, int lane, uint4 (&f)[8]) {
  TrRegs t;
  tr_issue<HI>(s, rb, lane, t);
  tr_collect<HI>(t, f);
}
// the 8 operand fragments (2 row blocks x 4 k-steps) of a 32-row group, all LDS reads issued together
__device__ __forceinline__ void group_frags(const bf16_t* s, int rb, int lane, uint4 (&f)[2][4]) {
#pragma unroll
  for (int c = 0; c < 2; ++c)
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) f[c][ks] = frag(s, rb + 16 * c, 32 * ks, lane);
}
// 4 consecutive head dims of one row -> 8-byte store; INV: conjugate RoPE (pairs p0, p0+1 of the row's tables)
template <bool INV>
__device__ __forceinline__ void store4(bf16_t* dst, const float (&x)[4], const float* c2, const float* s2) {
  float y[4] = {x[0], x[1], x[2], x[3]};
  if (INV) {
    const float2 c = *reinterpret_cast<const float2*>(c2), s = *reinterpret_cast<const float2*>(s2);
    y[0] = x[0] * c.x + x[1] * s.x; y[1] = x[1] * c.x - x[0] * s.x;
    y[2] = x[2] * c.y + x[3] * s.y; y[3] = x[3] * c.y - x[2] * s.y;
  }
  *reinterpret_cast<uint2*>(dst) = make_uint2(pack2(y[0], y[1]), pack2(y[2], y[3]));
}
// block-wide sum, blockDim.x == 512; `red` is 8 floats of LDS
__device__ __forceinline__ float block_sum_512(float v, float* red) {
  v = wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return ((red[0] + red[1]) + (red[2] + red[3])) + ((red[4] + red[5]) + (red[6] + red[7]));
}

// ------------------------------------------------------------------------------- forward
template <bool ROPE>
__global__ __launch_bounds__(512) void attn_fwd_mfma_k(const bf16_t* __restrict__ qkv, bf16_t* __restrict__ o,
                                                       float* __restrict__ lse_a, float* __restrict__ lse_t,
                                                       const float* __restrict__ gate1,
                                                       const float* __restrict__ gate2,
                                                       const int32_t* __restrict__ vstart,
                                                       const float* __restrict__ cs, const float* __restrict__ sn,
                                                       int n_seq, int S, int H, int A, int F) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  bf16_t* sK = reinterpret_cast<bf16_t*>(smem_raw);      // [BQ][LDR]  K rows (rotated)
  bf16_t* sV = sK + BQ * LDR;                             // [BQ][LDR]  V rows
  bf16_t* sKa = sV + BQ * LDR;                            // [16][LDR]  adapter K rows (zero beyond A)
  bf16_t* sVa = sKa + 16 * LDR;                           // [16][LDR]  adapter V rows
  // grid (H, n_seq, query blocks), LAST query block first: under the causal mask block qb walks qb + 1 key tiles, and the
  // dispatcher hands out workgroups in grid order — longest first keeps the short ones for filling the tail
  const int h = blockIdx.x, n = blockIdx.y, qb = (int)gridDim.z - 1 - (int)blockIdx.z;
  const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int g = lane >> 4, li = lane & 15;
  const int D = H * DH;
  const size_t ld = (size_t)3 * D;
  const bf16_t* seq = qkv + (size_t)n * S * ld + h * DH;
  const int i0 = qb * BQ + w * 16;                        // first query of this wave
  const int iq = i0 + li;                                 // this lane's query (score-block column)
  const int iqc = iq < S ? iq : S - 1;
  const float sc = rsqrtf((float)DH);

  // every global load of the first key tile goes out before the first wait
  uint4 qf[4];
  TileRegs<16> rKa, rVa;
  TileRegs<BQ> rK, rV;
  const bf16_t* arow = qkv + (size_t)n_seq * S * ld + h * DH;
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) qf[ks] = frag_g(seq, ld, iqc, 32 * ks, lane);
  tile_load<16>(rKa, arow + D, ld, 0, A);
  tile_load<16>(rVa, arow + 2 * D, ld, 0, A);
  tile_load<BQ>(rK, seq + D, ld, 0, S);
  tile_load<BQ>(rV, seq + 2 * D, ld, 0, S);
  if (ROPE) {
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
      qf[ks] = rope8(qf[ks], cs + (size_t)iqc * HP + 16 * ks + 4 * g, sn + (size_t)iqc * HP + 16 * ks + 4 * g);
  }
  tile_commit<false, 16>(rKa, sKa, 0, A, nullptr, nullptr);
  tile_commit<false, 16>(rVa, sVa, 0, A, nullptr, nullptr);

  f32x4 oacc[8];
#pragma unroll
  for (int d = 0; d < 8; ++d) oacc[d] = f32x4{0.f, 0.f, 0.f, 0.f};
  float m = NEG_BIG, ls = 0.f;                            // running max / this lane's part of the row sum
  const int vs = vstart[n];
  const float g2 = gate2[h];
  const bool biased_row = vs >= 0 && iq >= vs + F;
  constexpr float LOG2E = 1.44269504089f;
  const float sc2 = sc * LOG2E, g2l = g2 * LOG2E;        // scores in the log2 domain
  const bool bias_any = vs >= 0 && i0 + 15 >= vs + F;     // some row of this wave takes the frame bias (wave-uniform)

  AT_FWD_DECL
  for (int kt = 0; kt <= qb; ++kt) {
    if (kt > 0) __syncthreads();                          // tile kt-1 fully consumed
    AT_FWD_ADD(at_bar1);
    tile_commit<ROPE, BQ>(rK, sK, kt * BQ, S, cs, sn);
    tile_commit<false, BQ>(rV, sV, kt * BQ, S, nullptr, nullptr);
    if (kt < qb) {                                        // next tile's loads fly under this tile's arithmetic
      tile_load<BQ>(rK, seq + D, ld, (kt + 1) * BQ, S);
      tile_load<BQ>(rV, seq + 2 * D, ld, (kt + 1) * BQ, S);
    }
    AT_FWD_ADD(at_commit);
    __syncthreads();
    AT_FWD_ADD(at_bar2);
    const int jlast = min(i0 + 15, S - 1) - kt * BQ;      // last tile-local key any row of this wave sees
    // (a wave whose 16 queries all lie beyond S — the ragged last block — only takes part in the staging)
    const int ng = (jlast < 0 || i0 >= S) ? 0 : min(4, (jlast >> 5) + 1);
    for (int gq = 0; gq < ng; ++gq) {                     // 32 keys = two 16-key score blocks = one P·V k-step
      uint4 kfr[2][4];
      group_frags(sK, 32 * gq, lane, kfr);
      f32x4 st[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        st[0] = mma(kfr[0][ks], qf[ks], st[0]);
        st[1] = mma(kfr[1][ks], qf[ks], st[1]);
      }
      // The key loop is bound by the vector ALU (softmax bookkeeping), not by the matrix pipe: scores are kept in the
      // log2 domain (one FMA + one v_exp_f32 per element), and a group whose 32 keys lie below every query of the wave,
      // inside the sequence and outside the frame-bias window takes no mask / bias instructions at all (wave-uniform test).
      const int j0 = kt * BQ + 32 * gq;
      const bool edge = j0 + 31 > i0 || j0 + 31 >= S || (bias_any && j0 < vs + F && j0 + 31 >= vs);
      float v[2][4], mx = NEG_BIG;
      if (!edge) {
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            v[c][r] = st[c][r] * sc2;
            mx = fmaxf(mx, v[c][r]);
          }
      } else {
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int j = j0 + 16 * c + 4 * g + r;
            float x = st[c][r] * sc2;
            if (biased_row && j >= vs && j < vs + F) x += g2l;
            x = (j <= iq && j < S) ? x : NEG_BIG;
            v[c][r] = x;
            mx = fmaxf(mx, x);
          }
      }
      mx = across_g_max(mx);
      TrRegs tv;
      tr_issue<true>(sV, 32 * gq, lane, tv);              // V fragments return under the exponentials
      const float mn = fmaxf(m, mx);
      const float alpha = __builtin_amdgcn_exp2f(m - mn);
      m = mn;
      float p[2][4], rs = 0.f;
      const float mnc = mn > 0.5f * NEG_BIG ? mn : 0.f;   // a row with no key yet: every v is NEG_BIG, exp2 gives 0
#pragma unroll
      for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          p[c][r] = __builtin_amdgcn_exp2f(v[c][r] - mnc);
          rs += p[c][r];
        }
      ls = ls * alpha + rs;
      if (__any(alpha != 1.f)) {                          // the running maximum moved for some row of this wave
#pragma unroll
        for (int d = 0; d < 8; ++d)
#pragma unroll
          for (int r = 0; r < 4; ++r) oacc[d][r] *= alpha;
      }
      const uint4 pf = pack_blocks(p[0], p[1]);
      uint4 vf[8];
      tr_collect<true>(tv, vf);
#pragma unroll
      for (int d = 0; d < 8; ++d) oacc[d] = mma(vf[d], pf, oacc[d]);
    }
    AT_FWD_ADD(at_groups);
  }
  AT_FWD_DUMP(qb + 1, qb);
  const float l = across_g_sum(ls);
  const float inv = 1.f / l;
  const float lt = m * 0.69314718056f + __logf(l);        // m is a base-2 exponent
#pragma unroll
  for (int d = 0; d < 8; ++d)
#pragma unroll
    for (int r = 0; r < 4; ++r) oacc[d][r] *= inv;

  // ---- adapter prefix block: rows a = 4g+r of the score block, own softmax, scaled by tanh(gate1)
  f32x4 sa = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) sa = mma(frag(sKa, 0, 32 * ks, lane), qf[ks], sa);
  const float g1 = tanhf(gate1[h]);
  float xa[4], mxa = NEG_BIG;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    xa[r] = (4 * g + r < A) ? sa[r] * sc : NEG_BIG;
    mxa = fmaxf(mxa, xa[r]);
  }
  mxa = across_g_max(mxa);
  float ea[4], suma = 0.f;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    ea[r] = (4 * g + r < A) ? __expf(xa[r] - mxa) : 0.f;
    suma += ea[r];
  }
  suma = across_g_sum(suma);
  const float la = mxa + __logf(suma);
  {
    const float sca = g1 / suma;
    const float pa[4] = {ea[0] * sca, ea[1] * sca, ea[2] * sca, ea[3] * sca};
    const uint4 pf = make_uint4(pack2(pa[0], pa[1]), pack2(pa[2], pa[3]), 0u, 0u);
    uint4 vf[8];
    frags_tr_perm<false>(sVa, 0, lane, vf);
#pragma unroll
    for (int d = 0; d < 8; ++d) oacc[d] = mma(vf[d], pf, oacc[d]);
  }
  if (iq < S) {
    bf16_t* orow = o + ((size_t)n * S + iq) * D + h * DH + 4 * g;
#pragma unroll
    for (int d = 0; d < 8; ++d) {
      const float x[4] = {oacc[d][0], oacc[d][1], oacc[d][2], oacc[d][3]};
      store4<false>(orow + 16 * d, x, nullptr, nullptr);
    }
    if (g == 0) {
      lse_a[((size_t)n * H + h) * S + iq] = la;
      lse_t[((size_t)n * H + h) * S + iq] = lt;
    }
  }
}

// ------------------------------------------------------------------------------- backward: dQ
template <bool ROPE, bool RIN = ROPE>   // ROPE: conjugate rotation of dq / dk at the store; RIN: q, k arrive raw and are rotated on load
__global__ __launch_bounds__(512) void attn_bwd_dq_mfma_k(
    const bf16_t* __restrict__ d_o, const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ o,
    const float* __restrict__ lse_a, const float* __restrict__ lse_t, const float* __restrict__ gate1,
    const float* __restrict__ gate2, const int32_t* __restrict__ vstart, const float* __restrict__ cs,
    const float* __restrict__ sn, bf16_t* __restrict__ dqkv, float* __restrict__ delta_a,
    float* __restrict__ delta_t, float* __restrict__ gate_part, int n_seq, int S, int H, int A, int F) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  bf16_t* sK = reinterpret_cast<bf16_t*>(smem_raw);      // [BQ][LDR] K rows (rotated): row reads for Sᵀ, column reads for Kᵀ·dS
  bf16_t* sV = sK + BQ * LDR;                             // [BQ][LDR] V rows
  bf16_t* sKa = sV + BQ * LDR;                            // [16][LDR]
  bf16_t* sVa = sKa + 16 * LDR;                           // [16][LDR]
  __shared__ float red[16];
  // grid (H, n_seq, query blocks), LAST query block first: under the causal mask block qb walks qb + 1 key tiles, and the
  // dispatcher hands out workgroups in grid order — longest first keeps the short ones for filling the tail
  const int h = blockIdx.x, n = blockIdx.y, qb = (int)gridDim.z - 1 - (int)blockIdx.z;
  const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int g = lane >> 4, li = lane & 15;
  const int D = H * DH;
  const size_t ld = (size_t)3 * D;
  const bf16_t* seq = qkv + (size_t)n * S * ld + h * DH;
  const int i0 = qb * BQ + w * 16;
  const int iq = i0 + li;
  const int iqc = iq < S ? iq : S - 1;
  const bool live = iq < S;
  const float sc = rsqrtf((float)DH);
  const size_t sbase = ((size_t)n * H + h) * S;

  // every global load of the first key tile goes out before the first wait
  uint4 qf[4], dof[4], off[4];
  TileRegs<16> rKa, rVa;
  TileRegs<BQ> rK, rV;
  const bf16_t* arow = qkv + (size_t)n_seq * S * ld + h * DH;
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) {
    qf[ks] = frag_g(seq, ld, iqc, 32 * ks, lane);
    dof[ks] = frag_g(d_o + (size_t)n * S * D + h * DH, (size_t)D, iqc, 32 * ks, lane);
    off[ks] = frag_g(o + (size_t)n * S * D + h * DH, (size_t)D, iqc, 32 * ks, lane);
  }
  const float lsa = lse_a[sbase + iqc], lst = lse_t[sbase + iqc];
  tile_load<16>(rKa, arow + D, ld, 0, A);
  tile_load<16>(rVa, arow + 2 * D, ld, 0, A);
  tile_load<BQ>(rK, seq + D, ld, 0, S);
  tile_load<BQ>(rV, seq + 2 * D, ld, 0, S);
  float dtot = 0.f;                                       // dO·O over the row: this lane's 32 dims, then across g
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) {
    if (RIN) qf[ks] = rope8(qf[ks], cs + (size_t)iqc * HP + 16 * ks + 4 * g, sn + (size_t)iqc * HP + 16 * ks + 4 * g);
    const uint4 of = off[ks];
    const unsigned tw[4] = {dof[ks].x, dof[ks].y, dof[ks].z, dof[ks].w}, uw[4] = {of.x, of.y, of.z, of.w};
#pragma unroll
    for (int k = 0; k < 4; ++k)
      dtot += h16_lo(tw[k]) * h16_lo(uw[k]) +
              h16_hi(tw[k]) * h16_hi(uw[k]);
  }
  dtot = across_g_sum(dtot);
  const float g1 = tanhf(gate1[h]);
  const float g2 = gate2[h];
  const int vs = vstart[n];
  const bool biased_row = vs >= 0 && iq >= vs + F;
  const bool bias_any = vs >= 0 && i0 + 15 >= vs + F;     // some row of this wave takes the frame bias (wave-uniform)
  const float sc2 = sc * 1.44269504089f, lst2 = lst * 1.44269504089f, g2l = g2 * 1.44269504089f;   // log2 domain
  f32x4 dq[8];
#pragma unroll
  for (int d = 0; d < 8; ++d) dq[d] = f32x4{0.f, 0.f, 0.f, 0.f};

  // ---- adapter block: dS_a, delta_a, d tanh-gate partial
  tile_commit<false, 16>(rKa, sKa, 0, A, nullptr, nullptr);
  tile_commit<false, 16>(rVa, sVa, 0, A, nullptr, nullptr);
  tile_commit<RIN, BQ>(rK, sK, 0, S, cs, sn);
  tile_commit<false, BQ>(rV, sV, 0, S, nullptr, nullptr);
  __syncthreads();
  float da, dg1 = 0.f, dg2 = 0.f;
  {
    f32x4 sa = f32x4{0.f, 0.f, 0.f, 0.f}, dpa = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      sa = mma(frag(sKa, 0, 32 * ks, lane), qf[ks], sa);
      dpa = mma(frag(sVa, 0, 32 * ks, lane), dof[ks], dpa);
    }
    float pa[4], dov[4], part = 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const bool ok = 4 * g + r < A;
      pa[r] = ok ? __expf(sa[r] * sc - lsa) : 0.f;
      dov[r] = ok ? dpa[r] : 0.f;
      if (live) dg1 += dov[r] * pa[r];
      part += pa[r] * g1 * dov[r];
    }
    da = across_g_sum(part);
    const float ds[4] = {pa[0] * (g1 * dov[0] - da), pa[1] * (g1 * dov[1] - da), pa[2] * (g1 * dov[2] - da),
                         pa[3] * (g1 * dov[3] - da)};
    const uint4 sf = make_uint4(pack2(ds[0], ds[1]), pack2(ds[2], ds[3]), 0u, 0u);
    uint4 kf8[8];
    frags_tr_perm<false>(sKa, 0, lane, kf8);
#pragma unroll
    for (int d = 0; d < 8; ++d) dq[d] = mma(kf8[d], sf, dq[d]);
  }
  const float dt = dtot - da;

  if (qb > 0) {                                           // tile 1's loads fly under tile 0's arithmetic
    tile_load<BQ>(rK, seq + D, ld, BQ, S);
    tile_load<BQ>(rV, seq + 2 * D, ld, BQ, S);
  }
  for (int kt = 0; kt <= qb; ++kt) {
    if (kt > 0) {
      __syncthreads();                                    // tile kt-1 fully consumed
      tile_commit<RIN, BQ>(rK, sK, kt * BQ, S, cs, sn);
      tile_commit<false, BQ>(rV, sV, kt * BQ, S, nullptr, nullptr);
      if (kt < qb) {
        tile_load<BQ>(rK, seq + D, ld, (kt + 1) * BQ, S);
        tile_load<BQ>(rV, seq + 2 * D, ld, (kt + 1) * BQ, S);
      }
      __syncthreads();
    }
    const int jlast = min(i0 + 15, S - 1) - kt * BQ;
    // (a wave whose 16 queries all lie beyond S — the ragged last block — only takes part in the staging)
    const int ng = (jlast < 0 || i0 >= S) ? 0 : min(4, (jlast >> 5) + 1);
    for (int gq = 0; gq < ng; ++gq) {
      uint4 kfr[2][4], vfr[2][4];
      group_frags(sK, 32 * gq, lane, kfr);
      group_frags(sV, 32 * gq, lane, vfr);
      f32x4 st[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
      f32x4 dpt[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {                    // four independent accumulation chains
        st[0] = mma(kfr[0][ks], qf[ks], st[0]);
        st[1] = mma(kfr[1][ks], qf[ks], st[1]);
        dpt[0] = mma(vfr[0][ks], dof[ks], dpt[0]);
        dpt[1] = mma(vfr[1][ks], dof[ks], dpt[1]);
      }
      TrRegs tk;
      tr_issue<true>(sK, 32 * gq, lane, tk);              // Kᵀ fragments return under the exponentials
      float ds[2][4];
      // (the pass is bound by the vector ALU: a group whose 32 keys lie below every query of the wave, inside the sequence
      // and outside the frame-bias window takes one FMA + one v_exp_f32 per probability and no mask instructions)
      const int j0 = kt * BQ + 32 * gq;
      const bool edge = j0 + 31 > i0 || i0 + 15 >= S || (bias_any && j0 < vs + F && j0 + 31 >= vs);
      if (!edge) {
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
          for (int r = 0; r < 4; ++r)
            ds[c][r] = __builtin_amdgcn_exp2f(fmaf(st[c][r], sc2, -lst2)) * (dpt[c][r] - dt);
      } else {
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int j = j0 + 16 * c + 4 * g + r;
            float x = fmaf(st[c][r], sc2, -lst2);
            const bool inwin = biased_row && j >= vs && j < vs + F;
            if (inwin) x += g2l;
            const float p = (j <= iq && j < S && live) ? __builtin_amdgcn_exp2f(x) : 0.f;
            ds[c][r] = p * (dpt[c][r] - dt);
            if (inwin) dg2 += ds[c][r];
          }
      }
      const uint4 sf = pack_blocks(ds[0], ds[1]);
      uint4 kf8[8];
      tr_collect<true>(tk, kf8);
#pragma unroll
      for (int d = 0; d < 8; ++d) dq[d] = mma(kf8[d], sf, dq[d]);
    }
  }
  if (live) {
    bf16_t* row = dqkv + ((size_t)n * S + iq) * ld + h * DH + 4 * g;
    const float* cr = ROPE ? cs + (size_t)iq * HP + 2 * g : nullptr;
    const float* sr = ROPE ? sn + (size_t)iq * HP + 2 * g : nullptr;
#pragma unroll
    for (int d = 0; d < 8; ++d) {
      const float x[4] = {dq[d][0] * sc, dq[d][1] * sc, dq[d][2] * sc, dq[d][3] * sc};
      store4<ROPE>(row + 16 * d, x, cr + 8 * d, sr + 8 * d);
    }
    if (g == 0) {
      delta_a[sbase + iq] = da;
      delta_t[sbase + iq] = dt;
    }
  }
  const float b1 = block_sum_512(dg1, red);
  const float b2 = block_sum_512(dg2, red + 8);
  if (threadIdx.x == 0) {
    const size_t pidx = (((size_t)n * H + h) * gridDim.z + qb) * 2;
    gate_part[pidx] = b1;
    gate_part[pidx + 1] = b2;
  }
}

// ------------------------------------------------------------------------------- backward: dK, dV
// kb < nkb: 128 text keys (16 per wave); kb == nkb: the adapter keys (one 16-key block;
// waves 0-3 take the 32-query groups g ≡ w of every query tile and their partial sums meet in LDS).
template <bool ROPE, bool RIN = ROPE>   // ROPE: conjugate rotation of dq / dk at the store; RIN: q, k arrive raw and are rotated on load
__global__ __launch_bounds__(512) void attn_bwd_dkv_mfma_k(
    const bf16_t* __restrict__ d_o, const bf16_t* __restrict__ qkv, const float* __restrict__ lse_a,
    const float* __restrict__ lse_t, const float* __restrict__ delta_a, const float* __restrict__ delta_t,
    const float* __restrict__ gate1, const float* __restrict__ gate2, const int32_t* __restrict__ vstart,
    const float* __restrict__ cs, const float* __restrict__ sn, bf16_t* __restrict__ dqkv,
    float* __restrict__ dka_part, float* __restrict__ dva_part, int n_seq, int S, int H, int A, int F, int nkb) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  bf16_t* sQ = reinterpret_cast<bf16_t*>(smem_raw);      // [BQ][LDR] Q rows (rotated): row reads for S, column reads for Qᵀ·dS
  bf16_t* sdO = sQ + BQ * LDR;                            // [BQ][LDR] dO rows: row reads for dP, column reads for dOᵀ·P
  float* sL = reinterpret_cast<float*>(sdO + BQ * LDR);   // [BQ] lse, [BQ] delta of the staged queries
  float* sDl = sL + BQ;
  // grid (H, n_seq, key blocks + 1), longest first: the adapter block (kb == nkb: every query), then key block 0, 1, ...
  const int h = blockIdx.x, n = blockIdx.y, kb = blockIdx.z == 0 ? (int)gridDim.z - 1 : (int)blockIdx.z - 1;
  const bool adapter = kb == nkb;
  const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int g = lane >> 4, li = lane & 15;
  const int D = H * DH;
  const size_t ld = (size_t)3 * D;
  const bf16_t* seq = qkv + (size_t)n * S * ld + h * DH;
  const bf16_t* dob = d_o + (size_t)n * S * D + h * DH;
  const float sc = rsqrtf((float)DH);
  const size_t sbase = ((size_t)n * H + h) * S;
  f32x4 dk[8], dv[8];
#pragma unroll
  for (int d = 0; d < 8; ++d) { dk[d] = f32x4{0.f, 0.f, 0.f, 0.f}; dv[d] = f32x4{0.f, 0.f, 0.f, 0.f}; }
  const int nqt = (S + BQ - 1) / BQ;
  const float g1 = tanhf(gate1[h]);

  // this lane's key (score-block column) and its K / V operand fragments
  const int j0 = adapter ? 0 : kb * BQ + w * 16;
  const int jk = j0 + li;
  const int limit = adapter ? A : S;
  const bool kok = jk < limit;
  // every global load of the first query tile goes out before the first wait
  uint4 kf[4], vf[4];
  TileRegs<BQ> rQ, rdO;
  const int t_first = adapter ? 0 : kb;
  const int jc = kok ? jk : limit - 1;
  {
    const bf16_t* kbase = adapter ? qkv + (size_t)n_seq * S * ld + h * DH : seq;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      kf[ks] = frag_g(kbase + D, ld, jc, 32 * ks, lane);
      vf[ks] = frag_g(kbase + 2 * D, ld, jc, 32 * ks, lane);
    }
  }
  tile_load<BQ>(rQ, seq, ld, t_first * BQ, S);
  tile_load<BQ>(rdO, dob, (size_t)D, t_first * BQ, S);
  float l_in = 0.f, d_in = 0.f;
  if (threadIdx.x < BQ) {
    const int ii = min(t_first * BQ + (int)threadIdx.x, S - 1);
    l_in = adapter ? lse_a[sbase + ii] : lse_t[sbase + ii];
    d_in = adapter ? delta_a[sbase + ii] : delta_t[sbase + ii];
  }
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) {
    if (RIN && !adapter)
      kf[ks] = rope8(kf[ks], cs + (size_t)jc * HP + 16 * ks + 4 * g, sn + (size_t)jc * HP + 16 * ks + 4 * g);
    if (!kok) { kf[ks] = make_uint4(0, 0, 0, 0); vf[ks] = make_uint4(0, 0, 0, 0); }
  }
  const int vs = vstart[n];
  const float g2 = gate2[h];
  const bool win_key = vs >= 0 && jk >= vs && jk < vs + F;
  const bool win_any = vs >= 0 && j0 < vs + F && j0 + 15 >= vs;       // some key of this wave lies in the window (wave-uniform)
  constexpr float LOG2E = 1.44269504089f;
  const float sc2 = sc * LOG2E, g2l = g2 * LOG2E;
  for (int t = t_first; t < nqt; ++t) {
    if (t > t_first) __syncthreads();                     // tile t-1 fully consumed
    tile_commit<RIN, BQ>(rQ, sQ, t * BQ, S, cs, sn);
    tile_commit<false, BQ>(rdO, sdO, t * BQ, S, nullptr, nullptr);
    if (threadIdx.x < BQ) {
      sL[threadIdx.x] = l_in * LOG2E;                     // probabilities are formed in the log2 domain
      sDl[threadIdx.x] = d_in;
    }
    if (t + 1 < nqt) {                                    // next tile's loads fly under this tile's arithmetic
      tile_load<BQ>(rQ, seq, ld, (t + 1) * BQ, S);
      tile_load<BQ>(rdO, dob, (size_t)D, (t + 1) * BQ, S);
      if (threadIdx.x < BQ) {
        const int ii = min((t + 1) * BQ + (int)threadIdx.x, S - 1);
        l_in = adapter ? lse_a[sbase + ii] : lse_t[sbase + ii];
        d_in = adapter ? delta_a[sbase + ii] : delta_t[sbase + ii];
      }
    }
    __syncthreads();
    // 32-query groups of this tile that this wave works on
    int gbeg, gend, gstep;
    const int gmax = min(4, (min(S, (t + 1) * BQ) - t * BQ + 31) >> 5);          // groups holding real queries
    if (adapter) { gbeg = w; gend = w < 4 ? gmax : 0; gstep = 4; }
    else { gbeg = (t == kb) ? (w >> 1) : 0; gend = j0 < S ? gmax : 0; gstep = 1; }     // (no work for a wave of keys beyond S)
    for (int gq = gbeg; gq < gend; gq += gstep) {
      uint4 qfr[2][4], ofr[2][4];
      group_frags(sQ, 32 * gq, lane, qfr);
      group_frags(sdO, 32 * gq, lane, ofr);
      f32x4 s[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
      f32x4 dp[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {                    // four independent accumulation chains
        s[0] = mma(qfr[0][ks], kf[ks], s[0]);
        s[1] = mma(qfr[1][ks], kf[ks], s[1]);
        dp[0] = mma(ofr[0][ks], vf[ks], dp[0]);
        dp[1] = mma(ofr[1][ks], vf[ks], dp[1]);
      }
      TrRegs to, tq;
      tr_issue<true>(sdO, 32 * gq, lane, to);             // dOᵀ / Qᵀ fragments return under the exponentials
      tr_issue<true>(sQ, 32 * gq, lane, tq);
      float p[2][4], ds[2][4];
      // (vector-ALU bound: sL holds lse * log2(e), so a probability is one FMA + one v_exp_f32; a group whose 32 queries all
      // lie at or above every key of the wave, inside the sequence, for keys outside the frame-bias window, takes no masks)
      const int ig0 = t * BQ + 32 * gq;
      const bool edge = adapter || j0 + 15 > ig0 || ig0 + 31 >= S || j0 + 15 >= S || win_any;
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        const f32x4 lse4 = *reinterpret_cast<const f32x4*>(sL + 32 * gq + 16 * c + 4 * g);
        const f32x4 dl4 = *reinterpret_cast<const f32x4*>(sDl + 32 * gq + 16 * c + 4 * g);
        if (!edge) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            p[c][r] = __builtin_amdgcn_exp2f(fmaf(s[c][r], sc2, -lse4[r]));
            ds[c][r] = p[c][r] * (dp[c][r] - dl4[r]);
          }
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int i = ig0 + 16 * c + 4 * g + r;                 // query row of this register
            float x = fmaf(s[c][r], sc2, -lse4[r]);
            if (adapter) {
              const float pp = (kok && i < S) ? __builtin_amdgcn_exp2f(x) : 0.f;
              ds[c][r] = pp * (g1 * dp[c][r] - dl4[r]);
              p[c][r] = pp * g1;
            } else {
              if (win_key && i >= vs + F) x += g2l;
              const float pp = (jk <= i && i < S && kok) ? __builtin_amdgcn_exp2f(x) : 0.f;
              ds[c][r] = pp * (dp[c][r] - dl4[r]);
              p[c][r] = pp;
            }
          }
        }
      }
      const uint4 pf = pack_blocks(p[0], p[1]);
      const uint4 sf = pack_blocks(ds[0], ds[1]);
      uint4 t8[8];
      tr_collect<true>(to, t8);                           // waits for both sets (in-order return)
#pragma unroll
      for (int d = 0; d < 8; ++d) dv[d] = mma(t8[d], pf, dv[d]);
      tr_collect<true>(tq, t8);
#pragma unroll
      for (int d = 0; d < 8; ++d) dk[d] = mma(t8[d], sf, dk[d]);
    }
  }
  if (!adapter) {
    if (kok) {
      bf16_t* row = dqkv + ((size_t)n * S + jk) * ld + h * DH + 4 * g;
      const float* cr = ROPE ? cs + (size_t)jk * HP + 2 * g : nullptr;
      const float* sr = ROPE ? sn + (size_t)jk * HP + 2 * g : nullptr;
#pragma unroll
      for (int d = 0; d < 8; ++d) {
        const float xk[4] = {dk[d][0] * sc, dk[d][1] * sc, dk[d][2] * sc, dk[d][3] * sc};
        const float xv[4] = {dv[d][0], dv[d][1], dv[d][2], dv[d][3]};
        store4<ROPE>(row + D + 16 * d, xk, cr + 8 * d, sr + 8 * d);
        store4<false>(row + 2 * D + 16 * d, xv, nullptr, nullptr);
      }
    }
    return;
  }
  // adapter block: sum the four working waves' partial blocks through LDS ([4][16][128] fp32 x 2 = 64 KiB, the
  // Q/dO staging area is free now), then one fp32 partial per sequence for the batch reduction
  __syncthreads();
  float* rK = reinterpret_cast<float*>(smem_raw);          // [4][16][ADP_PITCH] fp32
  float* rV = rK + 4 * 16 * ADP_PITCH;
  if (w < 4) {
#pragma unroll
    for (int d = 0; d < 8; ++d) {
      *reinterpret_cast<f32x4*>(rK + (w * 16 + li) * ADP_PITCH + 16 * d + 4 * g) =
          f32x4{dk[d][0] * sc, dk[d][1] * sc, dk[d][2] * sc, dk[d][3] * sc};
      *reinterpret_cast<f32x4*>(rV + (w * 16 + li) * ADP_PITCH + 16 * d + 4 * g) = dv[d];
    }
  }
  __syncthreads();
  for (int idx = threadIdx.x; idx < A * DH; idx += 512) {
    const int aa = idx / DH, d = idx % DH;
    float sk = 0.f, sv = 0.f;
#pragma unroll
    for (int ww = 0; ww < 4; ++ww) {
      sk += rK[(ww * 16 + aa) * ADP_PITCH + d];
      sv += rV[(ww * 16 + aa) * ADP_PITCH + d];
    }
    const size_t oidx = ((size_t)n * A + aa) * D + h * DH + d;
    dka_part[oidx] = sk;
    dva_part[oidx] = sv;
  }
}


// ------------------------------------------------------------------------------- backward, S <= 128: one kernel
// One workgroup per (sequence, head) keeps Q, K, V, dO of the whole sequence in LDS (4 x 34 KiB) and runs the
// three passes back to back: (A) wave = 16 queries -> row deltas (barrier), dQ; (B) wave = 16 keys -> dK, dV; (C) waves 4-7
// -> the adapter keys' dK, dV (one 32-query group each, summed through LDS into this sequence's fp32 partial).
// The batch reduction of the adapter partials and of the gate sums is done by the LAST workgroup of each head to
// arrive (integer arrival counter in the workspace, self-resetting; partials are summed in sequence order, so the
// result does not depend on which workgroup is last): no separate reduction launch.
template <bool ROPE, bool RIN = ROPE>   // ROPE: conjugate rotation of dq / dk at the store; RIN: q, k arrive raw and are rotated on load
__global__ __launch_bounds__(512) void attn_bwd_fused_k(
    const bf16_t* __restrict__ d_o, const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ o,
    const float* __restrict__ lse_a, const float* __restrict__ lse_t, const float* __restrict__ gate1,
    const float* __restrict__ gate2, const int32_t* __restrict__ vstart, const float* __restrict__ cs,
    const float* __restrict__ sn, bf16_t* __restrict__ dqkv, float* __restrict__ dgate1, float* __restrict__ dgate2,
    float* gate_part, float* dka_part, float* dva_part, int* arrive,
    int n_seq, int S, int H, int A, int F) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  bf16_t* sQ = reinterpret_cast<bf16_t*>(smem_raw);      // [BQ][LDR] rotated Q
  bf16_t* sK = sQ + BQ * LDR;                             // [BQ][LDR] rotated K
  bf16_t* sV = sK + BQ * LDR;
  bf16_t* sdO = sV + BQ * LDR;
  bf16_t* sKa = sdO + BQ * LDR;                           // [16][LDR]
  bf16_t* sVa = sKa + 16 * LDR;
  float* sLt = reinterpret_cast<float*>(sVa + 16 * LDR);  // [BQ] lse of the causal softmax, adapter softmax,
  float* sLa = sLt + BQ;                                  //      then the row deltas written by pass A
  float* sDt = sLa + BQ;
  float* sDa = sDt + BQ;
  __shared__ float red[16];
  __shared__ int last_flag;
  const int h = blockIdx.x, n = blockIdx.y;
  AT_STAMP(0);
  const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int g = lane >> 4, li = lane & 15;
  const int D = H * DH;
  const size_t ld = (size_t)3 * D;
  const bf16_t* seq = qkv + (size_t)n * S * ld + h * DH;
  const bf16_t* dob = d_o + (size_t)n * S * D + h * DH;
  const bf16_t* arow = qkv + (size_t)n_seq * S * ld + h * DH;
  const float sc = rsqrtf((float)DH);
  const size_t sbase = ((size_t)n * H + h) * S;
  const int r16 = w * 16 + li;                            // this lane's query (pass A) / key (pass B)
  const int r16c = r16 < S ? r16 : S - 1;
  const bool live = r16 < S;
  const int gmax = min(4, (S + 31) >> 5);                 // 32-row groups holding real rows

  // ---- pass 0: every global load in flight, then rotate + stage
  uint4 off[4];
  {
    TileRegs<BQ> rQ, rK, rV, rdO;
    TileRegs<16> rKa, rVa;
    tile_load<BQ>(rQ, seq, ld, 0, S);
    tile_load<BQ>(rK, seq + D, ld, 0, S);
    tile_load<BQ>(rV, seq + 2 * D, ld, 0, S);
    tile_load<BQ>(rdO, dob, (size_t)D, 0, S);
    tile_load<16>(rKa, arow + D, ld, 0, A);
    tile_load<16>(rVa, arow + 2 * D, ld, 0, A);
    float lt_in = 0.f, la_in = 0.f;
    if (threadIdx.x < BQ) {
      const int ii = min((int)threadIdx.x, S - 1);
      lt_in = lse_t[sbase + ii];
      la_in = lse_a[sbase + ii];
    }
    // (the forward's output rows for the row deltas: requested with everything else, so that their latency runs under the
    // commits instead of in front of pass A)
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) off[ks] = frag_g(o + (size_t)n * S * D + h * DH, (size_t)D, r16c, 32 * ks, lane);
    tile_commit<RIN, BQ>(rQ, sQ, 0, S, cs, sn);
    tile_commit<RIN, BQ>(rK, sK, 0, S, cs, sn);
    tile_commit<false, BQ>(rV, sV, 0, S, nullptr, nullptr);
    tile_commit<false, BQ>(rdO, sdO, 0, S, nullptr, nullptr);
    tile_commit<false, 16>(rKa, sKa, 0, A, nullptr, nullptr);
    tile_commit<false, 16>(rVa, sVa, 0, A, nullptr, nullptr);
    if (threadIdx.x < BQ) {
      sLt[threadIdx.x] = lt_in;
      sLa[threadIdx.x] = la_in;
    }
  }
  const float g1 = tanhf(gate1[h]);
  const float g2 = gate2[h];
  const int vs = vstart[n];
  __syncthreads();
  AT_STAMP(1);

  float dg1 = 0.f, dg2 = 0.f;
  // ---- pass A: dQ of queries 16w..16w+15 (this lane: query r16, score-block column)
  {
    uint4 qf[4], dof[4];
    float dtot = 0.f;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      qf[ks] = frag(sQ, 16 * w, 32 * ks, lane);
      dof[ks] = frag(sdO, 16 * w, 32 * ks, lane);
      const unsigned tw[4] = {dof[ks].x, dof[ks].y, dof[ks].z, dof[ks].w};
      const unsigned uw[4] = {off[ks].x, off[ks].y, off[ks].z, off[ks].w};
#pragma unroll
      for (int k = 0; k < 4; ++k)
        dtot += h16_lo(tw[k]) * h16_lo(uw[k]) +
                h16_hi(tw[k]) * h16_hi(uw[k]);
    }
    dtot = across_g_sum(dtot);
    const float lsa = sLa[r16], lst = sLt[r16];
    const bool biased_row = vs >= 0 && r16 >= vs + F;
    f32x4 dq[8];
#pragma unroll
    for (int d = 0; d < 8; ++d) dq[d] = f32x4{0.f, 0.f, 0.f, 0.f};
    float da;
    {
      f32x4 sa = f32x4{0.f, 0.f, 0.f, 0.f}, dpa = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        sa = mma(frag(sKa, 0, 32 * ks, lane), qf[ks], sa);
        dpa = mma(frag(sVa, 0, 32 * ks, lane), dof[ks], dpa);
      }
      float pa[4], dov[4], part = 0.f;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const bool ok = 4 * g + r < A;
        pa[r] = ok ? __expf(sa[r] * sc - lsa) : 0.f;
        dov[r] = ok ? dpa[r] : 0.f;
        if (live) dg1 += dov[r] * pa[r];
        part += pa[r] * g1 * dov[r];
      }
      da = across_g_sum(part);
      const float ds[4] = {pa[0] * (g1 * dov[0] - da), pa[1] * (g1 * dov[1] - da), pa[2] * (g1 * dov[2] - da),
                           pa[3] * (g1 * dov[3] - da)};
      const uint4 sf = make_uint4(pack2(ds[0], ds[1]), pack2(ds[2], ds[3]), 0u, 0u);
      uint4 kf8[8];
      frags_tr_perm<false>(sKa, 0, lane, kf8);
#pragma unroll
      for (int d = 0; d < 8; ++d) dq[d] = mma(kf8[d], sf, dq[d]);
    }
    const float dt = dtot - da;
    // the row deltas are all pass B needs from pass A: publish them now, so that every wave can run its share of
    // pass A (w/2+1 key groups) and of pass B (4 - w/2 query groups) back to back — 5 groups each, no idle tail
    if (g == 0) {
      sDa[r16] = da;
      sDt[r16] = dt;
    }
    __syncthreads();
    const int ng = min(gmax, (w >> 1) + 1);               // keys 0 .. 16w+15
    for (int gq = 0; gq < ng; ++gq) {
      uint4 kfr[2][4], vfr[2][4];
      group_frags(sK, 32 * gq, lane, kfr);
      group_frags(sV, 32 * gq, lane, vfr);
      f32x4 st[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
      f32x4 dpt[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {                    // four independent accumulation chains
        st[0] = mma(kfr[0][ks], qf[ks], st[0]);
        st[1] = mma(kfr[1][ks], qf[ks], st[1]);
        dpt[0] = mma(vfr[0][ks], dof[ks], dpt[0]);
        dpt[1] = mma(vfr[1][ks], dof[ks], dpt[1]);
      }
      TrRegs tk;
      tr_issue<true>(sK, 32 * gq, lane, tk);              // Kᵀ fragments return under the exponentials
      float ds[2][4];
#pragma unroll
      for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int j = 32 * gq + 16 * c + 4 * g + r;
          float x = st[c][r] * sc;
          const bool inwin = biased_row && j >= vs && j < vs + F;
          if (inwin) x += g2;
          const float p = (j <= r16 && j < S && live) ? __expf(x - lst) : 0.f;
          ds[c][r] = p * (dpt[c][r] - dt);
          if (inwin) dg2 += ds[c][r];
        }
      const uint4 sf = pack_blocks(ds[0], ds[1]);
      uint4 kf8[8];
      tr_collect<true>(tk, kf8);
#pragma unroll
      for (int d = 0; d < 8; ++d) dq[d] = mma(kf8[d], sf, dq[d]);
    }
    if (live) {
      bf16_t* row = dqkv + ((size_t)n * S + r16) * ld + h * DH + 4 * g;
      const float* cr = ROPE ? cs + (size_t)r16 * HP + 2 * g : nullptr;
      const float* sr = ROPE ? sn + (size_t)r16 * HP + 2 * g : nullptr;
#pragma unroll
      for (int d = 0; d < 8; ++d) {
        const float x[4] = {dq[d][0] * sc, dq[d][1] * sc, dq[d][2] * sc, dq[d][3] * sc};
        store4<ROPE>(row + 16 * d, x, cr + 8 * d, sr + 8 * d);
      }
    }
  }

  AT_STAMP(2);
  // ---- pass B: dK, dV of keys 16w..16w+15 (this lane: key r16; rows >= S are zero in LDS)
  f32x4 dk[8], dv[8];
#pragma unroll
  for (int d = 0; d < 8; ++d) { dk[d] = f32x4{0.f, 0.f, 0.f, 0.f}; dv[d] = f32x4{0.f, 0.f, 0.f, 0.f}; }
  {
    uint4 kf[4], vf[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      kf[ks] = frag(sK, 16 * w, 32 * ks, lane);
      vf[ks] = frag(sV, 16 * w, 32 * ks, lane);
    }
    const bool win_key = vs >= 0 && r16 >= vs && r16 < vs + F;
    for (int gq = w >> 1; gq < gmax; ++gq) {
      uint4 qfr[2][4], ofr[2][4];
      group_frags(sQ, 32 * gq, lane, qfr);
      group_frags(sdO, 32 * gq, lane, ofr);
      f32x4 s[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
      f32x4 dp[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {                    // four independent accumulation chains
        s[0] = mma(qfr[0][ks], kf[ks], s[0]);
        s[1] = mma(qfr[1][ks], kf[ks], s[1]);
        dp[0] = mma(ofr[0][ks], vf[ks], dp[0]);
        dp[1] = mma(ofr[1][ks], vf[ks], dp[1]);
      }
      TrRegs to, tq;
      tr_issue<true>(sdO, 32 * gq, lane, to);             // dOᵀ / Qᵀ fragments return under the exponentials
      tr_issue<true>(sQ, 32 * gq, lane, tq);
      float p[2][4], ds[2][4];
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        const f32x4 lse4 = *reinterpret_cast<const f32x4*>(sLt + 32 * gq + 16 * c + 4 * g);
        const f32x4 dl4 = *reinterpret_cast<const f32x4*>(sDt + 32 * gq + 16 * c + 4 * g);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int i = 32 * gq + 16 * c + 4 * g + r;
          float x = s[c][r] * sc;
          if (win_key && i >= vs + F) x += g2;
          const float pp = (r16 <= i && i < S && live) ? __expf(x - lse4[r]) : 0.f;
          ds[c][r] = pp * (dp[c][r] - dl4[r]);
          p[c][r] = pp;
        }
      }
      const uint4 pf = pack_blocks(p[0], p[1]);
      const uint4 sf = pack_blocks(ds[0], ds[1]);
      uint4 t8[8];
      tr_collect<true>(to, t8);                           // waits for both sets (in-order return)
#pragma unroll
      for (int d = 0; d < 8; ++d) dv[d] = mma(t8[d], pf, dv[d]);
      tr_collect<true>(tq, t8);
#pragma unroll
      for (int d = 0; d < 8; ++d) dk[d] = mma(t8[d], sf, dk[d]);
    }
    if (live) {
      bf16_t* row = dqkv + ((size_t)n * S + r16) * ld + h * DH + 4 * g;
      const float* cr = ROPE ? cs + (size_t)r16 * HP + 2 * g : nullptr;
      const float* sr = ROPE ? sn + (size_t)r16 * HP + 2 * g : nullptr;
#pragma unroll
      for (int d = 0; d < 8; ++d) {
        const float xk[4] = {dk[d][0] * sc, dk[d][1] * sc, dk[d][2] * sc, dk[d][3] * sc};
        const float xv[4] = {dv[d][0], dv[d][1], dv[d][2], dv[d][3]};
        store4<ROPE>(row + D + 16 * d, xk, cr + 8 * d, sr + 8 * d);
        store4<false>(row + 2 * D + 16 * d, xv, nullptr, nullptr);
      }
    }
  }

  AT_STAMP(3);
  // ---- pass C: adapter keys (this lane: adapter row li), waves 4-7 take query group w-4
#pragma unroll
  for (int d = 0; d < 8; ++d) { dk[d] = f32x4{0.f, 0.f, 0.f, 0.f}; dv[d] = f32x4{0.f, 0.f, 0.f, 0.f}; }
  if (w >= 4 && w - 4 < gmax) {
    const int gq = w - 4;
    const bool aok = li < A;
    uint4 kf[4], vf[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      kf[ks] = frag(sKa, 0, 32 * ks, lane);
      vf[ks] = frag(sVa, 0, 32 * ks, lane);
    }
    uint4 qfr[2][4], ofr[2][4];
    group_frags(sQ, 32 * gq, lane, qfr);
    group_frags(sdO, 32 * gq, lane, ofr);
    f32x4 s[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
    f32x4 dp[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      s[0] = mma(qfr[0][ks], kf[ks], s[0]);
      s[1] = mma(qfr[1][ks], kf[ks], s[1]);
      dp[0] = mma(ofr[0][ks], vf[ks], dp[0]);
      dp[1] = mma(ofr[1][ks], vf[ks], dp[1]);
    }
    TrRegs to, tq;
    tr_issue<true>(sdO, 32 * gq, lane, to);
    tr_issue<true>(sQ, 32 * gq, lane, tq);
    float p[2][4], ds[2][4];
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      const f32x4 lse4 = *reinterpret_cast<const f32x4*>(sLa + 32 * gq + 16 * c + 4 * g);
      const f32x4 dl4 = *reinterpret_cast<const f32x4*>(sDa + 32 * gq + 16 * c + 4 * g);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int i = 32 * gq + 16 * c + 4 * g + r;
        const float pp = (aok && i < S) ? __expf(s[c][r] * sc - lse4[r]) : 0.f;
        ds[c][r] = pp * (g1 * dp[c][r] - dl4[r]);
        p[c][r] = pp * g1;
      }
    }
    const uint4 pf = pack_blocks(p[0], p[1]);
    const uint4 sf = pack_blocks(ds[0], ds[1]);
    uint4 t8[8];
    tr_collect<true>(to, t8);                           // waits for both sets (in-order return)
#pragma unroll
    for (int d = 0; d < 8; ++d) dv[d] = mma(t8[d], pf, dv[d]);
    tr_collect<true>(tq, t8);
#pragma unroll
    for (int d = 0; d < 8; ++d) dk[d] = mma(t8[d], sf, dk[d]);
  }
  __syncthreads();                                        // pass B is done with sK / sV: reuse them
  AT_STAMP(4);
  float* rK = reinterpret_cast<float*>(sK);               // [4][16][ADP_PITCH] fp32 (33 KiB <= one tile)
  float* rV = reinterpret_cast<float*>(sV);
  if (w >= 4) {
#pragma unroll
    for (int d = 0; d < 8; ++d) {
      *reinterpret_cast<f32x4*>(rK + ((w - 4) * 16 + li) * ADP_PITCH + 16 * d + 4 * g) =
          f32x4{dk[d][0] * sc, dk[d][1] * sc, dk[d][2] * sc, dk[d][3] * sc};
      *reinterpret_cast<f32x4*>(rV + ((w - 4) * 16 + li) * ADP_PITCH + 16 * d + 4 * g) = dv[d];
    }
  }
  __syncthreads();
  // Batch reduction by the LAST workgroup of this head to arrive. Cross-XCD visibility without a device-scope
  // fence (it writes back / scans the whole L2 on this multi-XCD part: measured +36 us per launch):
  //  * the partials are WRITTEN with device-scope relaxed atomic stores (sc1: written through to memory);
  //    every thread waits for its own stores (workgroup-scope release = s_waitcnt vmcnt(0)), the barrier
  //    collects the workgroup, and only then one thread bumps the arrival counter (device-scope RMW);
  //  * the last arriver does ONE device-scope acquire fence (lane 0, then a barrier) and reads the adapter
  //    partials with ordinary (pipelined) loads — the recipe of cdna_hip_programming.md "Projection GEMM at
  //    M = 256" item 2 for write-through slabs; device-scope (sc1) loads of every slab element instead
  //    measured 40 us slower;
  //  * the gate partials share lines between heads, so they are read with device-scope (sc1) loads.
  for (int idx = threadIdx.x; idx < A * DH; idx += 512) {
    const int aa = idx / DH, d = idx % DH;
    float sk = 0.f, sv = 0.f;
#pragma unroll
    for (int ww = 0; ww < 4; ++ww) {
      sk += rK[(ww * 16 + aa) * ADP_PITCH + d];
      sv += rV[(ww * 16 + aa) * ADP_PITCH + d];
    }
    const size_t oidx = ((size_t)n * A + aa) * D + h * DH + d;
    __hip_atomic_store(dka_part + oidx, sk, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(dva_part + oidx, sv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  const float b1 = block_sum_512(dg1, red);
  const float b2 = block_sum_512(dg2, red + 8);
  if (threadIdx.x == 0) {
    __hip_atomic_store(gate_part + ((size_t)n * H + h) * 2, b1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(gate_part + ((size_t)n * H + h) * 2 + 1, b2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // every wave: its write-through stores have completed
  __syncthreads();
  AT_STAMP(5);
  if (threadIdx.x == 0) {
    const int old = __hip_atomic_fetch_add(arrive + h, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    last_flag = (old == n_seq - 1) ? 1 : 0;
    if (last_flag) {                                      // the reducer: ONE device-scope acquire on behalf of the
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");  // workgroup before its ordinary loads of the other slabs
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
  }
  __syncthreads();
  AT_STAMP(6);
  if (!last_flag) return;
  for (int idx = threadIdx.x; idx < A * DH; idx += 512) {
    const int aa = idx / DH, d = idx % DH;
    float sk = 0.f, sv = 0.f;
    for (int n0 = 0; n0 < n_seq; n0 += 8) {               // 16 coherent loads in flight, summed in sequence order
      float tk[8], tv[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int nn = min(n0 + u, n_seq - 1);
        const size_t pi = ((size_t)nn * A + aa) * D + h * DH + d;
        tk[u] = dka_part[pi];
        tv[u] = dva_part[pi];
      }
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (n0 + u < n_seq) { sk += tk[u]; sv += tv[u]; }
    }
    bf16_t* row = dqkv + ((size_t)n_seq * S + aa) * ld + h * DH + d;
    row[0] = from_f32<bf16_t>(0.f);
    row[D] = from_f32<bf16_t>(sk);
    row[2 * D] = from_f32<bf16_t>(sv);
  }
  // gate sums: one coherent load per thread (all in flight together), then a fixed-order sum by thread 0
  float s1 = 0.f, s2 = 0.f;
  for (int n0 = 0; n0 < n_seq; n0 += BQ) {
    __syncthreads();
    if ((int)threadIdx.x < min(BQ, n_seq - n0)) {
      const size_t gi = ((size_t)(n0 + threadIdx.x) * H + h) * 2;
      sLt[threadIdx.x] = __hip_atomic_load(gate_part + gi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      sLa[threadIdx.x] = __hip_atomic_load(gate_part + gi + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    if (threadIdx.x == 0)
      for (int u = 0; u < min(BQ, n_seq - n0); ++u) { s1 += sLt[u]; s2 += sLa[u]; }
  }
  if (threadIdx.x == 0) {
    dgate1[h] += s1 * (1.f - g1 * g1);
    dgate2[h] += s2;
    __hip_atomic_store(arrive + h, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  AT_STAMP(7);
}

constexpr size_t FWD_LDS = (size_t)(2 * BQ + 32) * LDR * 2;
constexpr size_t FUSED_LDS = (size_t)(4 * BQ + 32) * LDR * 2 + 4 * BQ * 4;
constexpr size_t DKV_LDS = (size_t)2 * BQ * LDR * 2 + 2 * BQ * 4;
static_assert(DKV_LDS >= (size_t)2 * 4 * 16 * ADP_PITCH * 4, "adapter reduction reuses the staging area");
static_assert((size_t)BQ * LDR * 2 >= (size_t)4 * 16 * ADP_PITCH * 4, "a partial block fits the tile it reuses (fused backward)");

template <typename K>
void allow_lds(K kernel, size_t bytes) {
  (void)hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
}

}  // namespace

#ifdef FVQA_ATTN_STAMPS
extern "C" int fvqa_attn_stamps_read(unsigned long long* host, int clear) {
  if (hipMemcpyFromSymbol(host, HIP_SYMBOL(g_attn_stamps), sizeof(g_attn_stamps)) != hipSuccess) return -1;
  if (clear) {
    static unsigned long long zeros[1024 * 16];
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_attn_stamps), zeros, sizeof(zeros)) != hipSuccess) return -1;
  }
  return 1024 * 16;
}
#endif

// launched from attn.hip's C entry points when dtype == bf16 (workspace layout shared with the vector build).
// cos_t/sin_t != NULL: q and k in `qkv` are the raw projections and RoPE is applied on the fly.
int fvqa_attn_mfma_qblocks(int S) { return (S + BQ - 1) / BQ; }

int fvqa_attn_fwd_mfma(const void* qkv, void* o, float* lse_a, float* lse_t, const float* gate1, const float* gate2,
                       const int32_t* vstart, const float* cos_t, const float* sin_t, int n_seq, int S, int H, int A,
                       int F, hipStream_t st) {
  const int nqb = fvqa_attn_mfma_qblocks(S);
  static bool attr = false;
  if (!attr) {
    allow_lds(attn_fwd_mfma_k<true>, FWD_LDS);
    allow_lds(attn_fwd_mfma_k<false>, FWD_LDS);
    attr = true;
  }
  if (cos_t)
    hipLaunchKernelGGL(attn_fwd_mfma_k<true>, dim3(H, n_seq, nqb), dim3(512), FWD_LDS, st, (const bf16_t*)qkv,
                       (bf16_t*)o, lse_a, lse_t, gate1, gate2, vstart, cos_t, sin_t, n_seq, S, H, A, F);
  else
    hipLaunchKernelGGL(attn_fwd_mfma_k<false>, dim3(H, n_seq, nqb), dim3(512), FWD_LDS, st, (const bf16_t*)qkv,
                       (bf16_t*)o, lse_a, lse_t, gate1, gate2, vstart, cos_t, sin_t, n_seq, S, H, A, F);
  return 0;
}

// Returns 1 when the batch reduction (adapter rows of dqkv, dgate1/dgate2) has been done by the launch itself
// (S <= 128: fused kernel), 0 when the caller still has to run attn_bwd_reduce_k over nqb = fvqa_attn_mfma_qblocks(S).
int fvqa_attn_bwd_mfma(const void* d_o, const void* qkv, const void* o, const float* lse_a, const float* lse_t,
                       const float* gate1, const float* gate2, const int32_t* vstart, const float* cos_t,
                       const float* sin_t, void* dqkv, float* dgate1, float* dgate2, float* delta_a, float* delta_t,
                       float* gate_part, float* dka, float* dva, int* arrive, int n_seq, int S, int H, int A, int F,
                       hipStream_t st, int prerotated) {
  // prerotated (with tables): q, k in `qkv` are already rotated (fvqa_gemm_nt_rope), dq / dk are conjugate-rotated at the
  // store so that dqkv holds the gradients of the RAW projections
  const int nqb = fvqa_attn_mfma_qblocks(S);
  static bool attr = false;
  if (!attr) {
    allow_lds((attn_bwd_dq_mfma_k<true, false>), FWD_LDS);
    allow_lds((attn_bwd_dkv_mfma_k<true, false>), DKV_LDS);
    allow_lds((attn_bwd_fused_k<true, false>), FUSED_LDS);
    allow_lds(attn_bwd_dq_mfma_k<true>, FWD_LDS);
    allow_lds(attn_bwd_dq_mfma_k<false>, FWD_LDS);
    allow_lds(attn_bwd_dkv_mfma_k<true>, DKV_LDS);
    allow_lds(attn_bwd_dkv_mfma_k<false>, DKV_LDS);
    allow_lds(attn_bwd_fused_k<true>, FUSED_LDS);
    allow_lds(attn_bwd_fused_k<false>, FUSED_LDS);
    attr = true;
  }
  static const bool no_fuse = [] { const char* e = getenv("FVQA_ATTN_BWD_SPLIT"); return e && e[0] == '1'; }();
#define COMMA ,
  if (nqb == 1 && !no_fuse) {
#define FVQA_FUSED(R)                                                                                                 \
  hipLaunchKernelGGL((attn_bwd_fused_k<R>), dim3(H, n_seq), dim3(512), FUSED_LDS, st, (const bf16_t*)d_o,                \
                     (const bf16_t*)qkv, (const bf16_t*)o, lse_a, lse_t, gate1, gate2, vstart, cos_t, sin_t,          \
                     (bf16_t*)dqkv, dgate1, dgate2, gate_part, dka, dva, arrive, n_seq, S, H, A, F);
    if (cos_t && prerotated) { FVQA_FUSED(true COMMA false) } else if (cos_t) { FVQA_FUSED(true) } else { FVQA_FUSED(false) }
#undef FVQA_FUSED
    return 1;
  }
#define FVQA_BWD(R)                                                                                                   \
  hipLaunchKernelGGL((attn_bwd_dq_mfma_k<R>), dim3(H, n_seq, nqb), dim3(512), FWD_LDS, st, (const bf16_t*)d_o,           \
                     (const bf16_t*)qkv, (const bf16_t*)o, lse_a, lse_t, gate1, gate2, vstart, cos_t, sin_t,          \
                     (bf16_t*)dqkv, delta_a, delta_t, gate_part, n_seq, S, H, A, F);                                  \
  hipLaunchKernelGGL((attn_bwd_dkv_mfma_k<R>), dim3(H, n_seq, nqb + 1), dim3(512), DKV_LDS, st, (const bf16_t*)d_o,      \
                     (const bf16_t*)qkv, lse_a, lse_t, delta_a, delta_t, gate1, gate2, vstart, cos_t, sin_t,          \
                     (bf16_t*)dqkv, dka, dva, n_seq, S, H, A, F, nqb);
  if (cos_t && prerotated) { FVQA_BWD(true COMMA false) } else if (cos_t) { FVQA_BWD(true) } else { FVQA_BWD(false) }
#undef FVQA_BWD
#undef COMMA
  return 0;
}
