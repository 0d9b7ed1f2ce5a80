// 256x256-tile projection GEMM for gfx950 with a 4-deep LDS-DMA ring and split-K.
//
// Why a second tile size: a 128x128 tile has to pull (128+128) x 128 B per K-step for 512 MFMA
// cycles per SIMD, i.e. 64 B/clk per CU — the CU's whole L2->LDS fill rate — so that family tops
// out near 700 TF/s on MI355X whatever its schedule. A 256x256 tile needs 32 B/clk.
//
// Structure (512 threads = 8 waves as 2(M) x 4(N), 128x64 outputs per wave = 8x4 MFMA 16x16 tiles,
// 128 accumulator registers):
//   * K is consumed in stages of 64 bytes per row (32 bf16 / 16 fp32) = one MFMA k-step;
//   * a stage is 16 KiB of A + 16 KiB of B, filled by 4 global_load_lds_dwordx4 per wave;
//   * the LDS ring holds 4 stages (128 KiB): stage t+3 is issued while stage t is computed, so a
//     load has three compute periods (~1.5k cycles) to land — enough to stream weights from HBM
//     with ONE workgroup per CU;
//   * one raw s_barrier per stage, guarded by a COUNTED s_waitcnt vmcnt (never 0 in steady state):
//     the barrier both publishes stage t (every wave's DMA for it has landed) and retires the
//     reads of stage t-1, whose slot the next DMA overwrites;
//   * LDS image is lane-linear (DMA constraint); the 64-byte rows are XOR-swizzled on the SOURCE
//     address and on the ds_read_b128 fragment read with g(row) = (-(row>>2)) & 3, which makes
//     every 16-lane read group hit 16 distinct 16-byte bank slots.
// Grid filling: M is only 1024..3072 rows here, so outputs with few tiles (N = 4096: 64 tiles on
// 256 CUs) are split along K over blockIdx ranges; partial sums go to an fp32 workspace
// [split][M][N] and `splitk_fixup` adds them (+ residual, fp32 tail rows) in one pass.
#include "common.h"
#include "probe.h"

namespace {

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int TM = 256, TN = 256;
constexpr int SROW = 64;                 // bytes of K per row per stage
constexpr int STAGE_OP = TM * SROW;      // 16 KiB per operand per stage
constexpr int STAGE = 2 * STAGE_OP;      // 32 KiB
constexpr int NSTAGE = 4;                // ring depth of the plain loop (PIPE == 0)
constexpr int LDS256 = NSTAGE * STAGE;   // 128 KiB

// run(a, b, acc) feeds the WEIGHT fragment as the MFMA's row operand and the activation fragment as
// its column operand, i.e. each 16x16 block is produced transposed: lane l then holds
// out[m = l&15][n = 4*(l>>4) .. +3] — four CONSECUTIVE output columns per lane, so the epilogue
// moves 8/16-byte vectors (C, partial sums, the SwiGLU' operands) instead of 2/4-byte scalars.
template <typename T> struct Mma256;
template <> struct Mma256<bf16_t> {
  static constexpr int KE = 32;          // elements per stage
  static __device__ __forceinline__ void run(const u32x4& a, const u32x4& b, f32x4& acc) {
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, b),
                                                  __builtin_bit_cast(bf16x8_t, a), acc, 0, 0, 0);
  }
};
template <> struct Mma256<float> {
  static constexpr int KE = 16;
  static __device__ __forceinline__ void run(const u32x4& a, const u32x4& b, f32x4& acc) {
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(b.x), __uint_as_float(a.x), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(b.y), __uint_as_float(a.y), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(b.z), __uint_as_float(a.z), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(b.w), __uint_as_float(a.w), acc, 0, 0, 0);
  }
};

__device__ __forceinline__ void unpack8_bf16(const uint4& q, float (&v)[8]) {
  v[0] = __uint_as_float(q.x << 16); v[1] = __uint_as_float(q.x & 0xFFFF0000u);
  v[2] = __uint_as_float(q.y << 16); v[3] = __uint_as_float(q.y & 0xFFFF0000u);
  v[4] = __uint_as_float(q.z << 16); v[5] = __uint_as_float(q.z & 0xFFFF0000u);
  v[6] = __uint_as_float(q.w << 16); v[7] = __uint_as_float(q.w & 0xFFFF0000u);
}
template <typename TO> __device__ __forceinline__ void store8(TO* p, const float (&v)[8]);
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

__device__ __forceinline__ int xcd_remap256(int bid, int nwg) {
  const int q = nwg >> 3, r = nwg & 7, x = bid & 7;
  const int base = (x < r) ? x * (q + 1) : r * (q + 1) + (x - r) * q;
  return base + (bid >> 3);
}

// EPI: 0 none, 1 residual. SPLIT: write fp32 partials to `ws` instead of C.
// PIPE (loop structure): 0 = 64-byte rows, symmetric 4-slot ring (kept for ablation);
//       2 = 128-byte rows, 2-slot ring; 3 = 128-byte rows, asymmetric A/B rings with role-split DMA;
//       6 = 3 plus the half-stage stagger of waves 4-7 (default).
// NT: 16-column MFMA blocks per wave along N (4 -> 256-wide tile; 3 -> 192-wide, PIPE 6 only).
template <typename T, typename TO, int EPI, bool SPLIT, int PIPE, int NT = 4>
__global__ __launch_bounds__(512) void gemm_nt_256(const T* __restrict__ A, const T* __restrict__ B,
                                                   TO* __restrict__ C, const T* __restrict__ R,
                                                   float* __restrict__ tail, float* __restrict__ ws, int M, int N,
                                                   int K, int lda, int ldb, int ldc, int m_split, int tiles_m,
                                                   int splits) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int KE = Mma256<T>::KE;
  constexpr int CH = 16 / (int)sizeof(T);

  // work id -> (m tile fastest, then K split, then n tile): neighbours share the weight panel
  const int wid = xcd_remap256(blockIdx.x, gridDim.x);
  const int tmi = wid % tiles_m;
  const int sp = (wid / tiles_m) % splits;
  const int tni = wid / (tiles_m * splits);
  static_assert(NT == 4 || (NT == 3 && PIPE == 6), "192-wide tiles exist for the default loop only");
  constexpr int WN = 16 * NT;              // output columns per wave
#if defined(FVQA_ABLATE) && (FVQA_ABLATE & 32)
  const int m0 = 0, n0 = 0;      // timing experiment: every workgroup streams the same (L2-resident) tiles
#else
  const int m0 = tmi * TM, n0 = tni * (4 * WN);
#endif
  const int nk_all = K / KE;
  // K range of this split in stages; the wide-row loop eats stages in pairs, so its ranges are even
  constexpr int KU = (PIPE == 2 || PIPE == 3 || PIPE == 6) ? 2 : 1;
  const int kbeg = KU * (int)(((long long)(nk_all / KU) * sp) / splits);
  const int kend = KU * (int)(((long long)(nk_all / KU) * (sp + 1)) / splits);
  const int nk = kend - kbeg;

  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = w >> 2, wc = w & 3;

  // ---- DMA geometry: wave w fills pieces 2w, 2w+1 of A and of B (piece = 16 rows x 64 B = 1 KiB)
  const T* srcA[2];
  const T* srcB[2];
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const int piece = w * 2 + t;
#if defined(FVQA_ABLATE) && (FVQA_ABLATE & 16)
    // timing experiment only (results are garbage): every DMA instruction fetches whole 128-B lines
    const int row = piece * 8 + (lane >> 3);
    const int c = (lane & 7);
#else
    const int row = piece * 16 + (lane >> 2);
    const int c = (lane & 3) ^ ((-(row >> 2)) & 3);
#endif
    int ga = m0 + row; ga = ga < M ? ga : M - 1;
    int gb = n0 + row; gb = gb < N ? gb : N - 1;
    srcA[t] = A + (size_t)ga * lda + (size_t)kbeg * KE + c * CH;
    srcB[t] = B + (size_t)gb * ldb + (size_t)kbeg * KE + c * CH;
  }
  auto issue_slot = [&](int st, int slot) {  // stage index st (relative to kbeg) -> ring slot
    char* dA = smem + slot * STAGE;
    char* dB = dA + STAGE_OP;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int piece = w * 2 + t;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(srcA[t] + (size_t)st * KE),
                                       (__attribute__((address_space(3))) void*)(dA + piece * 1024), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(srcB[t] + (size_t)st * KE),
                                       (__attribute__((address_space(3))) void*)(dB + piece * 1024), 16, 0, 0);
    }
  };
  auto issue = [&](int st) { issue_slot(st, st & (NSTAGE - 1)); };

  f32x4 acc[8][NT];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // fragment read: row (lane&15) of each 16-row block, 16-byte chunk (lane>>4), swizzled
  const int frow = lane & 15;
  const int fch = ((lane >> 4) ^ ((-(frow >> 2)) & 3)) << 4;
  const int offA = (wr * 128 + frow) * SROW + fch;
  const int offB = STAGE_OP + (wc * 64 + frow) * SROW + fch;

  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
  // LDS fragment reads are inline asm on purpose: hipcc's waitcnt pass makes every compiler-visible
  // LDS load wait for ALL in-flight LDS-DMA (s_waitcnt vmcnt(0)), which would drain the ring each
  // stage. The reads return in issue order, so counted lgkmcnt waits release the MFMAs; each wait
  // statement names the registers it guards ("+v") so no consumer is scheduled above it.
#if defined(FVQA_ABLATE) && (FVQA_ABLATE & 4)
#define FVQA_DSR(dst, addr, off) dst = u32x4{(unsigned)(addr), (unsigned)(off), 0x3f803f80u, 0x3f803f80u}
#else
#define FVQA_DSR(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "i"(off))
#endif
#define FVQA_READ12(A_, B_, slot)                                                                          \
  {                                                                                                          \
    const unsigned sb_ = lds0 + (unsigned)((slot) * STAGE);                                                  \
    const unsigned pa_ = sb_ + (unsigned)offA, pb_ = sb_ + (unsigned)offB;                                   \
    FVQA_DSR(B_[0], pb_, 0);    FVQA_DSR(B_[1], pb_, 1024); FVQA_DSR(B_[2], pb_, 2048); FVQA_DSR(B_[3], pb_, 3072); \
    FVQA_DSR(A_[0], pa_, 0);    FVQA_DSR(A_[1], pa_, 1024); FVQA_DSR(A_[2], pa_, 2048); FVQA_DSR(A_[3], pa_, 3072); \
    FVQA_DSR(A_[4], pa_, 4096); FVQA_DSR(A_[5], pa_, 5120); FVQA_DSR(A_[6], pa_, 6144); FVQA_DSR(A_[7], pa_, 7168); \
  }
  auto wait_dma = [&](int newer) {           // my DMA groups newer than the awaited one: 4 loads each
    if (newer >= 3) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    else if (newer == 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if (newer == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  };

  if constexpr (PIPE == 6) {
    // Mode 3 plus a half-stage STAGGER: waves 4-7 (the second wave on every SIMD) run k-step 1 of
    // each stage after the next stage's barrier, from fragments they read (and waited for) before it.
    // While waves 0-3 sit in the LDS latency that follows a barrier, waves 4-7 have 32 MFMAs queued,
    // so the matrix pipe does not drain at every stage boundary (MI355X_MICROARCH 'Two waves per SIMD' item 9).
    // ---- wide-row asymmetric rings. Measured on MI355X (profiles/r01_gemm_ablation.log): with every
    // workgroup streaming the same L2-resident tiles the DMA path alone runs 1.8 PF/s-equivalent, on
    // the real operands 1.2 — the ring was waiting on HBM latency of the once-read WEIGHT panel, not
    // on bandwidth. So the weight (B) operand gets a deeper ring than the activations (A, re-read
    // by every N tile, served by L2 / Infinity Cache): A 2 x 32 KiB, B 3 x 32 KiB = 160 KiB.
    // vmcnt retires in order per wave, so the two streams are issued by DIFFERENT waves: waves 0-3
    // move A and wait vmcnt(0) for stage u; waves 4-7 move B two stages ahead and wait with one
    // newer stage (8 loads) still in flight. One barrier per stage publishes both.
    constexpr int WROW = 128, WOP = TM * WROW;            // A: 32 KiB per stage
    constexpr int WOPB = 4 * WN * WROW;                   // B: 32 KiB (NT 4) / 24 KiB (NT 3) per stage
    constexpr int NPB = 2 * NT;                           // 1-KiB pieces per B wave per stage
    const int nw = nk / 2;
    const bool bwave = w >= 4;                            // wave-uniform DMA role
    const int wq = w & 3;
    const T* wsrc[8];
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      // pieces of 8 rows x 128 B: 32 per A stage (8 per wave), 8*NT per B stage (NPB per wave)
      const int piece = bwave ? wq * NPB + (t < NPB ? t : NPB - 1) : wq * 8 + t;
      const int row = piece * 8 + (lane >> 3);
      const int c = (lane & 7) ^ (row & 7);
      int ga = m0 + row; ga = ga < M ? ga : M - 1;
      int gb = n0 + row; gb = gb < N ? gb : N - 1;
      wsrc[t] = bwave ? B + (size_t)gb * ldb + (size_t)kbeg * KE + c * CH
                      : A + (size_t)ga * lda + (size_t)kbeg * KE + c * CH;
    }
    char* const ringA = smem;                             // 2 slots
    char* const ringB = smem + 2 * WOP;                   // 3 slots
    // piece q (0..7) of this wave's operand for stage u; slot = ring slot of that stage
    auto issue_piece = [&](int u, int slot, int q) {
      char* d = bwave ? ringB + slot * WOPB + (wq * NPB + q) * 1024 : ringA + slot * WOP + (wq * 8 + q) * 1024;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(wsrc[q] + (size_t)u * 2 * KE),
                                       (__attribute__((address_space(3))) void*)d, 16, 0, 0);
    };
    const int fsw = lane & 7, fkc = lane >> 4;
    const unsigned rowA = (unsigned)((wr * 128 + frow) * WROW);
    const unsigned rowB = (unsigned)(2 * WOP + (wc * WN + frow) * WROW);
    const unsigned ck0 = (unsigned)(((0 + fkc) ^ fsw) << 4), ck1 = (unsigned)(((4 + fkc) ^ fsw) << 4);
#define FVQA_WREAD(A_, B_, pa0, pb0, ck)                                                                     \
  {                                                                                                          \
    const unsigned pa_ = (pa0) + (ck), pb_ = (pb0) + (ck);                                                   \
    FVQA_DSR(B_[0], pb_, 0);    FVQA_DSR(B_[1], pb_, 2048);  FVQA_DSR(B_[2], pb_, 4096);                               \
    if constexpr (NT == 4) FVQA_DSR(B_[3], pb_, 6144);                                                       \
    FVQA_DSR(A_[0], pa_, 0);    FVQA_DSR(A_[1], pa_, 2048);  FVQA_DSR(A_[2], pa_, 4096);  FVQA_DSR(A_[3], pa_, 6144);  \
    FVQA_DSR(A_[4], pa_, 8192); FVQA_DSR(A_[5], pa_, 10240); FVQA_DSR(A_[6], pa_, 12288); FVQA_DSR(A_[7], pa_, 14336); \
  }
#if defined(FVQA_ABLATE) && (FVQA_ABLATE & 1)
#define FVQA_WROW(i, n)                                                                \
  asm volatile("s_waitcnt lgkmcnt(" #n ")" : "+v"(a[i]));                              \
  asm volatile("" ::"v"(a[i]), "v"(b[0]), "v"(b[1]), "v"(b[2]));                       \
  __builtin_amdgcn_sched_barrier(0);
#else
#define FVQA_WROW(i, n)                                                                \
  asm volatile("s_waitcnt lgkmcnt(" #n ")" : "+v"(a[i]));                              \
  Mma256<T>::run(a[i], b[0], acc[i][0]); Mma256<T>::run(a[i], b[1], acc[i][1]);        \
  Mma256<T>::run(a[i], b[2], acc[i][2]);                                               \
  if constexpr (NT == 4) Mma256<T>::run(a[i], b[3], acc[i][NT - 1]);                   \
  __builtin_amdgcn_sched_barrier(0);
#endif
#if defined(FVQA_ABLATE) && (FVQA_ABLATE & 2)
#define FVQA_WROW_DMA(i, n, q) FVQA_WROW(i, n)
#else
#if defined(FVQA_AEARLY) && FVQA_AEARLY == 1
#define FVQA_WROW_DMA(i, n, q)                                                         \
  FVQA_WROW(i, n)                                                                      \
  if (more) {                                                                          \
    if (bwave) { if (q < NPB) issue_piece(nu, nslot, q); }                             \
    else if (q < 4) { issue_piece(nu, nslot, 2 * q); issue_piece(nu, nslot, 2 * q + 1); } \
  }                                                                                    \
  __builtin_amdgcn_sched_barrier(0);
#elif defined(FVQA_AEARLY) && FVQA_AEARLY == 2
#define FVQA_WROW_DMA(i, n, q)                                                         \
  FVQA_WROW(i, n)                                                                      \
  if (more && bwave && q < NPB) issue_piece(nu, nslot, q);                             \
  __builtin_amdgcn_sched_barrier(0);
#else
#define FVQA_WROW_DMA(i, n, q)                                                         \
  FVQA_WROW(i, n)                                                                      \
  if (more && (!bwave || q < NPB)) issue_piece(nu, nslot, q);                          \
  __builtin_amdgcn_sched_barrier(0);
#endif
#endif
    // prologue: A stage 0; B stages 0 and 1
    if (nw > 0) {
#pragma unroll
      for (int q = 0; q < 8; ++q)
        if (!bwave || q < NPB) issue_piece(0, 0, q);
      if (bwave && nw > 1) {
#pragma unroll
        for (int q = 0; q < NPB; ++q) issue_piece(1, 1, q);
      }
    }
    int sa = 0, sbs = 0;                                  // ring slots of stage u: u % 2, u % 3
    const bool late = w >= 4;                             // wave-uniform
    // (a static s_setprio 1 for either half of the waves measured -5 % / 0 %: the stagger already orders them)
    // k-step-1 fragments of output rows 0-63 (B + first four A rows): consumed at once by waves 0-3,
    // held across the next barrier by waves 4-7 (16 deferred MFMAs = 256 cycles of cover)
    u32x4 hl[4], hb[4];
#define FVQA_HROW(i)                                                                   \
  Mma256<T>::run(hl[i], hb[0], acc[i][0]); Mma256<T>::run(hl[i], hb[1], acc[i][1]);    \
  Mma256<T>::run(hl[i], hb[2], acc[i][2]);                                             \
  if constexpr (NT == 4) Mma256<T>::run(hl[i], hb[3], acc[i][NT - 1]);
#define FVQA_UROW(i)                                                                   \
  Mma256<T>::run(au[i - 4], hb[0], acc[i][0]); Mma256<T>::run(au[i - 4], hb[1], acc[i][1]); \
  Mma256<T>::run(au[i - 4], hb[2], acc[i][2]);                                              \
  if constexpr (NT == 4) Mma256<T>::run(au[i - 4], hb[3], acc[i][NT - 1]);
    for (int u = 0; u < nw; ++u) {
      if (bwave && u + 1 < nw) {                          // B(u) landed, B(u+1) (NPB loads) in flight
        if constexpr (NT == 4) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
      }
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();                       // stage u published; slots of stage u-1 are free
      asm volatile("" ::: "memory");
      const int nu = bwave ? u + 2 : u + 1;
      const int nslot = bwave ? (sbs == 0 ? 2 : sbs - 1) : (sa ^ 1);
      const bool more = nu < nw;
      const unsigned pa0 = lds0 + (unsigned)(sa * WOP) + rowA;
      const unsigned pb0 = lds0 + (unsigned)(sbs * WOPB) + rowB;
#if defined(FVQA_AEARLY) && FVQA_AEARLY == 2
      if (more && !bwave) {
#pragma unroll
        for (int q = 0; q < 8; ++q) issue_piece(nu, nslot, q);
        __builtin_amdgcn_sched_barrier(0);
      }
#endif
      if (late && u > 0) {                                // deferred rows 0-63 of stage u-1, k-step 1
        FVQA_HROW(0) FVQA_HROW(1) FVQA_HROW(2) FVQA_HROW(3)
        __builtin_amdgcn_sched_barrier(0);
      }
      {
        u32x4 a[8], b[4];
        FVQA_WREAD(a, b, pa0, pb0, ck0);
        asm volatile("s_waitcnt lgkmcnt(7)" : "+v"(b[0]), "+v"(b[1]), "+v"(b[2]));
        if constexpr (NT == 4) asm volatile("" : "+v"(b[3]));
        FVQA_WROW_DMA(0, 7, 0) FVQA_WROW_DMA(1, 6, 1) FVQA_WROW_DMA(2, 5, 2) FVQA_WROW_DMA(3, 4, 3)
        FVQA_WROW_DMA(4, 3, 4) FVQA_WROW_DMA(5, 2, 5) FVQA_WROW_DMA(6, 1, 6) FVQA_WROW_DMA(7, 0, 7)
      }
      {
        u32x4 au[4];
        const unsigned pa_ = pa0 + ck1, pb_ = pb0 + ck1;
        FVQA_DSR(hb[0], pb_, 0);    FVQA_DSR(hb[1], pb_, 2048);  FVQA_DSR(hb[2], pb_, 4096);
        if constexpr (NT == 4) FVQA_DSR(hb[3], pb_, 6144);
        FVQA_DSR(au[0], pa_, 8192); FVQA_DSR(au[1], pa_, 10240); FVQA_DSR(au[2], pa_, 12288); FVQA_DSR(au[3], pa_, 14336);
        FVQA_DSR(hl[0], pa_, 0);    FVQA_DSR(hl[1], pa_, 2048);  FVQA_DSR(hl[2], pa_, 4096);  FVQA_DSR(hl[3], pa_, 6144);
        asm volatile("s_waitcnt lgkmcnt(4)"
                     : "+v"(hb[0]), "+v"(hb[1]), "+v"(hb[2]), "+v"(au[0]), "+v"(au[1]), "+v"(au[2]), "+v"(au[3]));
        if constexpr (NT == 4) asm volatile("" : "+v"(hb[3]));
        FVQA_UROW(4) FVQA_UROW(5) FVQA_UROW(6) FVQA_UROW(7)
        __builtin_amdgcn_sched_barrier(0);
        // retire the reads of rows 0-63 before the next barrier (their slot may be refilled after it)
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(hl[0]), "+v"(hl[1]), "+v"(hl[2]), "+v"(hl[3]));
        if (!late) {
          FVQA_HROW(0) FVQA_HROW(1) FVQA_HROW(2) FVQA_HROW(3)
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      asm volatile("" ::: "memory");
      sa ^= 1;
      sbs = (sbs == 2) ? 0 : sbs + 1;
    }
    if (late && nw > 0) {
      FVQA_HROW(0) FVQA_HROW(1) FVQA_HROW(2) FVQA_HROW(3)
    }
#undef FVQA_UROW
#undef FVQA_HROW
#undef FVQA_WROW_DMA
#undef FVQA_WROW
#undef FVQA_WREAD
  } else
  if constexpr (PIPE == 3) {
    // ---- wide-row asymmetric rings. Measured on MI355X (profiles/r01_gemm_ablation.log): with every
    // workgroup streaming the same L2-resident tiles the DMA path alone runs 1.8 PF/s-equivalent, on
    // the real operands 1.2 — the ring was waiting on HBM latency of the once-read WEIGHT panel, not
    // on bandwidth. So the weight (B) operand gets a deeper ring than the activations (A, re-read
    // by every N tile, served by L2 / Infinity Cache): A 2 x 32 KiB, B 3 x 32 KiB = 160 KiB.
    // vmcnt retires in order per wave, so the two streams are issued by DIFFERENT waves: waves 0-3
    // move A and wait vmcnt(0) for stage u; waves 4-7 move B two stages ahead and wait with one
    // newer stage (8 loads) still in flight. One barrier per stage publishes both.
    constexpr int WROW = 128, WOP = TM * WROW;            // 32 KiB per operand per stage
    const int nw = nk / 2;
    const bool bwave = w >= 4;                            // wave-uniform DMA role
    const int wq = w & 3;
    const T* wsrc[8];
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      const int piece = wq * 8 + t;                       // 32 pieces of 8 rows x 128 B per operand
      const int row = piece * 8 + (lane >> 3);
      const int c = (lane & 7) ^ (row & 7);
      int ga = m0 + row; ga = ga < M ? ga : M - 1;
      int gb = n0 + row; gb = gb < N ? gb : N - 1;
      wsrc[t] = bwave ? B + (size_t)gb * ldb + (size_t)kbeg * KE + c * CH
                      : A + (size_t)ga * lda + (size_t)kbeg * KE + c * CH;
    }
    char* const ringA = smem;                             // 2 slots
    char* const ringB = smem + 2 * WOP;                   // 3 slots
    // piece q (0..7) of this wave's operand for stage u; slot = ring slot of that stage
    auto issue_piece = [&](int u, int slot, int q) {
      char* d = (bwave ? ringB : ringA) + slot * WOP + (wq * 8 + q) * 1024;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(wsrc[q] + (size_t)u * 2 * KE),
                                       (__attribute__((address_space(3))) void*)d, 16, 0, 0);
    };
    const int fsw = lane & 7, fkc = lane >> 4;
    const unsigned rowA = (unsigned)((wr * 128 + frow) * WROW);
    const unsigned rowB = (unsigned)(2 * WOP + (wc * 64 + frow) * WROW);
    const unsigned ck0 = (unsigned)(((0 + fkc) ^ fsw) << 4), ck1 = (unsigned)(((4 + fkc) ^ fsw) << 4);
#define FVQA_WREAD(A_, B_, pa0, pb0, ck)                                                                     \
  {                                                                                                          \
    const unsigned pa_ = (pa0) + (ck), pb_ = (pb0) + (ck);                                                   \
    FVQA_DSR(B_[0], pb_, 0);    FVQA_DSR(B_[1], pb_, 2048);  FVQA_DSR(B_[2], pb_, 4096);  FVQA_DSR(B_[3], pb_, 6144);  \
    FVQA_DSR(A_[0], pa_, 0);    FVQA_DSR(A_[1], pa_, 2048);  FVQA_DSR(A_[2], pa_, 4096);  FVQA_DSR(A_[3], pa_, 6144);  \
    FVQA_DSR(A_[4], pa_, 8192); FVQA_DSR(A_[5], pa_, 10240); FVQA_DSR(A_[6], pa_, 12288); FVQA_DSR(A_[7], pa_, 14336); \
  }
#if defined(FVQA_ABLATE) && (FVQA_ABLATE & 1)
#define FVQA_WROW(i, n)                                                                \
  asm volatile("s_waitcnt lgkmcnt(" #n ")" : "+v"(a[i]));                              \
  asm volatile("" ::"v"(a[i]), "v"(b[0]), "v"(b[1]), "v"(b[2]), "v"(b[3]));            \
  __builtin_amdgcn_sched_barrier(0);
#else
#define FVQA_WROW(i, n)                                                                \
  asm volatile("s_waitcnt lgkmcnt(" #n ")" : "+v"(a[i]));                              \
  Mma256<T>::run(a[i], b[0], acc[i][0]); Mma256<T>::run(a[i], b[1], acc[i][1]);        \
  Mma256<T>::run(a[i], b[2], acc[i][2]); Mma256<T>::run(a[i], b[3], acc[i][3]);        \
  __builtin_amdgcn_sched_barrier(0);
#endif
#if defined(FVQA_ABLATE) && (FVQA_ABLATE & 2)
#define FVQA_WROW_DMA(i, n, q) FVQA_WROW(i, n)
#else
#define FVQA_WROW_DMA(i, n, q)                                                         \
  FVQA_WROW(i, n)                                                                      \
  if (more) issue_piece(nu, nslot, q);                                                 \
  __builtin_amdgcn_sched_barrier(0);
#endif
    // prologue: A stage 0; B stages 0 and 1
    if (nw > 0) {
#pragma unroll
      for (int q = 0; q < 8; ++q) issue_piece(0, 0, q);
      if (bwave && nw > 1) {
#pragma unroll
        for (int q = 0; q < 8; ++q) issue_piece(1, 1, q);
      }
    }
    int sa = 0, sbs = 0;                                  // ring slots of stage u: u % 2, u % 3
    for (int u = 0; u < nw; ++u) {
      if (bwave && u + 1 < nw) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");   // B(u) landed, B(u+1) in flight
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#if !(defined(FVQA_ABLATE) && (FVQA_ABLATE & 8))
      __builtin_amdgcn_s_barrier();                       // stage u published; slots of stage u-1 are free
#endif
      asm volatile("" ::: "memory");
      // next DMA of this wave: A(u+1) -> A slot (u+1)%2, or B(u+2) -> B slot (u+2)%3
      const int nu = bwave ? u + 2 : u + 1;
      const int nslot = bwave ? (sbs == 0 ? 2 : sbs - 1) : (sa ^ 1);
      const bool more = nu < nw;
      const unsigned pa0 = lds0 + (unsigned)(sa * WOP) + rowA;
      const unsigned pb0 = lds0 + (unsigned)(sbs * WOP) + rowB;
      {
        u32x4 a[8], b[4];
        FVQA_WREAD(a, b, pa0, pb0, ck0);
        asm volatile("s_waitcnt lgkmcnt(7)" : "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]));
        FVQA_WROW_DMA(0, 7, 0) FVQA_WROW_DMA(1, 6, 1) FVQA_WROW_DMA(2, 5, 2) FVQA_WROW_DMA(3, 4, 3)
        FVQA_WROW_DMA(4, 3, 4) FVQA_WROW_DMA(5, 2, 5) FVQA_WROW_DMA(6, 1, 6) FVQA_WROW_DMA(7, 0, 7)
      }
      {
        u32x4 a[8], b[4];
        FVQA_WREAD(a, b, pa0, pb0, ck1);
        asm volatile("s_waitcnt lgkmcnt(7)" : "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]));
        FVQA_WROW(0, 7) FVQA_WROW(1, 6) FVQA_WROW(2, 5) FVQA_WROW(3, 4)
        FVQA_WROW(4, 3) FVQA_WROW(5, 2) FVQA_WROW(6, 1) FVQA_WROW(7, 0)
      }
      asm volatile("" ::: "memory");
      sa ^= 1;
      sbs = (sbs == 2) ? 0 : sbs + 1;
    }
#undef FVQA_WROW_DMA
#undef FVQA_WROW
#undef FVQA_WREAD
  } else
  if constexpr (PIPE == 2) {
    // ---- wide-row ring: a stage is 128 bytes of K per row (64 bf16 / 32 fp32), so every DMA lane
    // quad-pair fetches WHOLE 128-byte cache lines (the 64-byte-row loops issue one L2 request per
    // half line: the L2 request rate, not bytes, bounded them). Stage = 32 KiB A + 32 KiB B, ring of
    // two stages; each stage is consumed as two MFMA k-steps. Rows are XOR-swizzled by (row & 7)
    // over their eight 16-byte chunks (conflict-free for ds_read_b128, as in gemm_nt_128).
    constexpr int WROW = 128, WOP = TM * WROW, WSTAGE = 2 * WOP;
    const int nw = nk / 2;                       // wide stages in this K range (host guarantees even)
    const T* wA[4];
    const T* wB[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int piece = w * 4 + t;               // 32 pieces of 8 rows x 128 B per operand
      const int row = piece * 8 + (lane >> 3);
      const int c = (lane & 7) ^ (row & 7);
      int ga = m0 + row; ga = ga < M ? ga : M - 1;
      int gb = n0 + row; gb = gb < N ? gb : N - 1;
      wA[t] = A + (size_t)ga * lda + (size_t)kbeg * KE + c * CH;
      wB[t] = B + (size_t)gb * ldb + (size_t)kbeg * KE + c * CH;
    }
    // one DMA piece (1 KiB) of stage u: q = 0..3 -> A pieces, 4..7 -> B pieces of this wave
    auto issue_piece = [&](int u, int q) {
      char* d = smem + (u & 1) * WSTAGE + (q >> 2) * WOP + (w * 4 + (q & 3)) * 1024;
      const T* src = ((q >> 2) ? wB[q & 3] : wA[q & 3]) + (size_t)u * 2 * KE;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                       (__attribute__((address_space(3))) void*)d, 16, 0, 0);
    };
    auto issue_w = [&](int u) {
#pragma unroll
      for (int q = 0; q < 8; ++q) issue_piece(u, q);
    };
    const int fsw = lane & 7, fkc = lane >> 4;
    const unsigned rowA = (unsigned)((wr * 128 + frow) * WROW);
    const unsigned rowB = (unsigned)(WOP + (wc * 64 + frow) * WROW);
    const unsigned ck0 = (unsigned)(((0 + fkc) ^ fsw) << 4), ck1 = (unsigned)(((4 + fkc) ^ fsw) << 4);
#define FVQA_WREAD(A_, B_, sb, ck)                                                                           \
  {                                                                                                          \
    const unsigned pa_ = (sb) + rowA + (ck), pb_ = (sb) + rowB + (ck);                                       \
    FVQA_DSR(B_[0], pb_, 0);    FVQA_DSR(B_[1], pb_, 2048);  FVQA_DSR(B_[2], pb_, 4096);  FVQA_DSR(B_[3], pb_, 6144);  \
    FVQA_DSR(A_[0], pa_, 0);    FVQA_DSR(A_[1], pa_, 2048);  FVQA_DSR(A_[2], pa_, 4096);  FVQA_DSR(A_[3], pa_, 6144);  \
    FVQA_DSR(A_[4], pa_, 8192); FVQA_DSR(A_[5], pa_, 10240); FVQA_DSR(A_[6], pa_, 12288); FVQA_DSR(A_[7], pa_, 14336); \
  }
#if defined(FVQA_ABLATE) && (FVQA_ABLATE & 1)
#define FVQA_WROW(i, n)                                                                \
  asm volatile("s_waitcnt lgkmcnt(" #n ")" : "+v"(a[i]));                              \
  asm volatile("" ::"v"(a[i]), "v"(b[0]), "v"(b[1]), "v"(b[2]), "v"(b[3]));            \
  __builtin_amdgcn_sched_barrier(0);
#else
#define FVQA_WROW(i, n)                                                                \
  asm volatile("s_waitcnt lgkmcnt(" #n ")" : "+v"(a[i]));                              \
  Mma256<T>::run(a[i], b[0], acc[i][0]); Mma256<T>::run(a[i], b[1], acc[i][1]);        \
  Mma256<T>::run(a[i], b[2], acc[i][2]); Mma256<T>::run(a[i], b[3], acc[i][3]);        \
  __builtin_amdgcn_sched_barrier(0);
#endif
    // The eight DMA issues of the next stage are interleaved one per MFMA row of the first k-step
    // (an LDS-DMA issue costs ~100 cycles of the wave's issue stream; behind an MFMA row it hides).
#define FVQA_WROW_DMA(i, n, q)                                                         \
  FVQA_WROW(i, n)                                                                      \
  if (more) issue_piece(u + 1, q);                                                     \
  __builtin_amdgcn_sched_barrier(0);
    if (nw > 0) issue_w(0);
    for (int u = 0; u < nw; ++u) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // my DMA of stage u has landed
      __builtin_amdgcn_s_barrier();                          // ... and everyone's; slot (u+1)&1 is free
      asm volatile("" ::: "memory");
      const bool more = u + 1 < nw;
      const unsigned sb = lds0 + (unsigned)((u & 1) * WSTAGE);
      {
        u32x4 a[8], b[4];
        FVQA_WREAD(a, b, sb, ck0);
        asm volatile("s_waitcnt lgkmcnt(7)" : "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]));
        FVQA_WROW_DMA(0, 7, 0) FVQA_WROW_DMA(1, 6, 4) FVQA_WROW_DMA(2, 5, 1) FVQA_WROW_DMA(3, 4, 5)
        FVQA_WROW_DMA(4, 3, 2) FVQA_WROW_DMA(5, 2, 6) FVQA_WROW_DMA(6, 1, 3) FVQA_WROW_DMA(7, 0, 7)
      }
      {
        u32x4 a[8], b[4];
        FVQA_WREAD(a, b, sb, ck1);
        asm volatile("s_waitcnt lgkmcnt(7)" : "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]));
        FVQA_WROW(0, 7) FVQA_WROW(1, 6) FVQA_WROW(2, 5) FVQA_WROW(3, 4)
        FVQA_WROW(4, 3) FVQA_WROW(5, 2) FVQA_WROW(6, 1) FVQA_WROW(7, 0)
      }
      asm volatile("" ::: "memory");
    }
#undef FVQA_WROW_DMA
#undef FVQA_WROW
#undef FVQA_WREAD
  } else
  if constexpr (PIPE == 0) {
  if (nk > 0) issue(0);
  if (nk > 1) issue(1);
  if (nk > 2) issue(2);
#define FVQA_ROW(i, n)                                                                 \
  asm volatile("s_waitcnt lgkmcnt(" #n ")" : "+v"(a[i]));                              \
  Mma256<T>::run(a[i], b[0], acc[i][0]); Mma256<T>::run(a[i], b[1], acc[i][1]);        \
  Mma256<T>::run(a[i], b[2], acc[i][2]); Mma256<T>::run(a[i], b[3], acc[i][3]);        \
  __builtin_amdgcn_sched_barrier(0);
#ifndef FVQA_ABLATE
#define FVQA_ABLATE 0      /* tuning builds only: 1 no MFMA, 2 no in-loop DMA, 4 no LDS reads, 8 no barrier */
#endif
  for (int t = 0; t < nk; ++t) {
    if (!(FVQA_ABLATE & 2)) wait_dma(min(nk - t - 1, 2));
    if (!(FVQA_ABLATE & 8)) __builtin_amdgcn_s_barrier();   // stage t complete for all waves; slot (t-1)&3 free
    asm volatile("" ::: "memory");
    if (!(FVQA_ABLATE & 2) && t + 3 < nk) issue(t + 3);
    u32x4 a[8], b[4];
    if (FVQA_ABLATE & 4) {
#pragma unroll
      for (int i = 0; i < 8; ++i) a[i] = u32x4{(unsigned)t, 1u, 2u, 3u};
#pragma unroll
      for (int j = 0; j < 4; ++j) b[j] = u32x4{(unsigned)t, 5u, 6u, 7u};
    } else {
      FVQA_READ12(a, b, (t & (NSTAGE - 1)));
    }
    if (FVQA_ABLATE & 1) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int i = 0; i < 8; ++i) asm volatile("" ::"v"(a[i]));
#pragma unroll
      for (int j = 0; j < 4; ++j) asm volatile("" ::"v"(b[j]));
    } else {
      asm volatile("s_waitcnt lgkmcnt(7)" : "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]));
      FVQA_ROW(0, 7) FVQA_ROW(1, 6) FVQA_ROW(2, 5) FVQA_ROW(3, 4)
      FVQA_ROW(4, 3) FVQA_ROW(5, 2) FVQA_ROW(6, 1) FVQA_ROW(7, 0)
    }
    asm volatile("" ::: "memory");
  }
#undef FVQA_ROW
  }
#undef FVQA_READ12
#undef FVQA_DSR

  // ---- epilogue (transposed C/D map: m = lane&15, n = (lane>>4)*4 + reg; see Mma256)
  const int crow = lane & 15;
  const int ccol = (lane >> 4) * 4;
#if defined(FVQA_ABLATE) && (FVQA_ABLATE & 64)
  {                                    // timing experiment: no epilogue traffic (one never-taken store keeps acc live)
    float s_ = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j) s_ += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
    if (s_ == 1.2345678e-30f) ws[0] = s_;
    return;
  }
#endif
  // Wide path: the accumulators make one round trip through LDS (free once the ring has drained) so
  // that every store instruction covers whole 128-byte lines: 8 lanes per row, 16 bytes per lane
  // (measured: split-K partial stores 16 -> 10 us per launch against 4-column fragments).
  // Each wave stages its own 128 x 64 tile in two 64-row passes through a private 16 KiB region;
  // 16-byte chunk c of row r sits at chunk c ^ (r & 15), which keeps both the fragment writes (16 rows,
  // one chunk) and the row reads (2 rows x 8 chunks per 16 lanes) free of bank conflicts.
  //   4-byte outputs: lane k of a row owns columns 4k..4k+3 and 32+4k..32+4k+3 (two full lines per row);
  //   2-byte outputs: lane k owns columns 8k..8k+7 (one full line per row).
  const bool wide_ok = tail == nullptr && ((N & 7) == 0) && ((ldc & 7) == 0) &&
                       ((((uintptr_t)C | (uintptr_t)R | (uintptr_t)ws) & 15) == 0);
  if (wide_ok) {
    constexpr bool W4 = SPLIT || sizeof(TO) == 4;
    __syncthreads();                                   // every wave is done reading the ring
    float* stg = reinterpret_cast<float*>(smem) + w * (64 * 64);
    const int q = lane >> 3, k = lane & 7;
    const int rl = W4 ? ((q & 1) * 8 + (q >> 1)) : q;  // row of this lane inside an 8-row group set
    const int cA = W4 ? k : 2 * k, cB = W4 ? 8 + k : 2 * k + 1;
    const bool okA = cA < 4 * NT, okB = cB < 4 * NT;      // a 192-wide tile leaves chunks 12..15 of the row unused
    const int nA = okA ? n0 + wc * WN + cA * 4 : N, nB = okB ? n0 + wc * WN + cB * 4 : N;
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      const int mb = m0 + wr * 128 + p * 64;
      auto row_of = [&](int t) { return W4 ? 16 * (t >> 1) + 4 * (t & 1) + rl : 8 * t + rl; };
      uint4 qa[8], qb[8];
      if constexpr (!SPLIT && sizeof(T) == 2 && (EPI == FVQA_EPI_SWIGLU_BWD || EPI == FVQA_EPI_RESIDUAL)) {
#pragma unroll
        for (int t = 0; t < 8; ++t) {                  // epilogue operands in flight before the staging
          const int m = mb + row_of(t);
          qa[t] = qb[t] = uint4{0u, 0u, 0u, 0u};
          if (m < M && nA < N) {
            if constexpr (EPI == FVQA_EPI_SWIGLU_BWD) {
              const T* rp = R + (size_t)m * ldc + nA;        // ab rows: a | b halves, b at +ldc/2
              qa[t] = *reinterpret_cast<const uint4*>(rp);
              qb[t] = *reinterpret_cast<const uint4*>(rp + (ldc >> 1));
            } else {
              qa[t] = *reinterpret_cast<const uint4*>(R + (size_t)m * ldc + nA);
            }
          }
        }
      }
#pragma unroll
      for (int ii = 0; ii < 4; ++ii)
#pragma unroll
        for (int j = 0; j < NT; ++j)
          *reinterpret_cast<f32x4*>(stg + (ii * 16 + crow) * 64 + (((j * 4 + (lane >> 4)) ^ crow) << 2)) =
              acc[p * 4 + ii][j];
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int t = 0; t < 8; ++t) {
        const int r = row_of(t);
        const int m = mb + r;
        float v[8];
        {
          const f32x4 lo = *reinterpret_cast<const f32x4*>(stg + r * 64 + ((cA ^ (r & 15)) << 2));
          const f32x4 hi = *reinterpret_cast<const f32x4*>(stg + r * 64 + ((cB ^ (r & 15)) << 2));
          v[0] = lo[0]; v[1] = lo[1]; v[2] = lo[2]; v[3] = lo[3];
          v[4] = hi[0]; v[5] = hi[1]; v[6] = hi[2]; v[7] = hi[3];
        }
        if (m >= M) continue;
        float(&vA)[4] = reinterpret_cast<float(&)[4]>(v[0]);
        float(&vB)[4] = reinterpret_cast<float(&)[4]>(v[4]);
        if constexpr (SPLIT) {
          float* wp = ws + ((size_t)sp * M + m) * N;
          if (nA < N) Vec4<float>::store(wp + nA, vA);
          if (nB < N) Vec4<float>::store(wp + nB, vB);
        } else if constexpr (EPI == FVQA_EPI_SWIGLU_BWD) {
          const size_t o = (size_t)m * ldc;
          const int hb_ = ldc >> 1;
          float a_[8], b_[8], da[8], db[8];
          if constexpr (sizeof(T) == 2) {
            if (nA >= N) continue;
            unpack8_bf16(qa[t], a_);
            unpack8_bf16(qb[t], b_);
          } else {
            if (nA < N) { Vec4<T>::load(R + o + nA, reinterpret_cast<float(&)[4]>(a_[0]));
                          Vec4<T>::load(R + o + hb_ + nA, reinterpret_cast<float(&)[4]>(b_[0])); }
            if (nB < N) { Vec4<T>::load(R + o + nB, reinterpret_cast<float(&)[4]>(a_[4]));
                          Vec4<T>::load(R + o + hb_ + nB, reinterpret_cast<float(&)[4]>(b_[4])); }
          }
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const float sg = 1.f / (1.f + __expf(-a_[e]));
            da[e] = v[e] * b_[e] * sg * (1.f + a_[e] * (1.f - sg));
            db[e] = v[e] * a_[e] * sg;
          }
          if constexpr (sizeof(TO) == 2) {
            store8<TO>(C + o + nA, da);
            store8<TO>(C + o + hb_ + nA, db);
          } else {
            if (nA < N) { Vec4<TO>::store(C + o + nA, reinterpret_cast<float(&)[4]>(da[0]));
                          Vec4<TO>::store(C + o + hb_ + nA, reinterpret_cast<float(&)[4]>(db[0])); }
            if (nB < N) { Vec4<TO>::store(C + o + nB, reinterpret_cast<float(&)[4]>(da[4]));
                          Vec4<TO>::store(C + o + hb_ + nB, reinterpret_cast<float(&)[4]>(db[4])); }
          }
        } else {
          TO* cp = C + (size_t)m * ldc;
          if constexpr (EPI == FVQA_EPI_RESIDUAL) {
            float r_[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            if constexpr (sizeof(T) == 2) {
              unpack8_bf16(qa[t], r_);
            } else {
              if (nA < N) Vec4<T>::load(R + (size_t)m * ldc + nA, reinterpret_cast<float(&)[4]>(r_[0]));
              if (nB < N) Vec4<T>::load(R + (size_t)m * ldc + nB, reinterpret_cast<float(&)[4]>(r_[4]));
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] += r_[e];
          }
          if constexpr (sizeof(TO) == 2) {
            if (nA < N) store8<TO>(cp + nA, v);
          } else {
            if (nA < N) Vec4<TO>::store(cp + nA, vA);
            if (nB < N) Vec4<TO>::store(cp + nB, vB);
          }
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      __builtin_amdgcn_wave_barrier();
    }
    return;
  }
  const bool vec_ok = ((N & 3) == 0) && ((ldc & 3) == 0) &&
                      ((((uintptr_t)C | (uintptr_t)R | (uintptr_t)ws | (uintptr_t)tail) & 15) == 0);
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int m = m0 + wr * 128 + i * 16 + crow;
    if (m >= M) continue;
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const int n = n0 + wc * WN + j * 16 + ccol;
      if (n >= N) continue;
      float v[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
      if (vec_ok) {                    // n % 4 == 0 and N % 4 == 0: the four columns are all inside
        if (SPLIT) {
          Vec4<float>::store(ws + ((size_t)sp * M + m) * N + n, v);
        } else if (tail != nullptr && m >= m_split) {
          float* tp = tail + (size_t)(m - m_split) * N + n;
          float u[4];
          Vec4<float>::load(tp, u);
          u[0] += v[0]; u[1] += v[1]; u[2] += v[2]; u[3] += v[3];
          Vec4<float>::store(tp, u);
        } else if (EPI == FVQA_EPI_SWIGLU_BWD) {
          // v = dz[m][n..n+3]; R = ab (rows of 2N: a | b); C = dab (rows of 2N): d(silu(a)*b)
          const size_t o = (size_t)m * ldc + n;
          float a_[4], b_[4], da[4], db[4];
          Vec4<T>::load(R + o, a_);
          Vec4<T>::load(R + o + (ldc >> 1), b_);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float sg = 1.f / (1.f + __expf(-a_[e]));
            da[e] = v[e] * b_[e] * sg * (1.f + a_[e] * (1.f - sg));
            db[e] = v[e] * a_[e] * sg;
          }
          Vec4<TO>::store(C + o, da);
          Vec4<TO>::store(C + o + (ldc >> 1), db);
        } else {
          if (EPI == FVQA_EPI_RESIDUAL) {
            float r[4];
            Vec4<T>::load(R + (size_t)m * ldc + n, r);
            v[0] += r[0]; v[1] += r[1]; v[2] += r[2]; v[3] += r[3];
          }
          Vec4<TO>::store(C + (size_t)m * ldc + n, v);
        }
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          if (n + e >= N) continue;
          if (SPLIT) {
            ws[((size_t)sp * M + m) * N + n + e] = v[e];
          } else if (tail != nullptr && m >= m_split) {
            tail[(size_t)(m - m_split) * N + n + e] += v[e];
          } else if (EPI == FVQA_EPI_SWIGLU_BWD) {
            const size_t o = (size_t)m * ldc + n + e;
            const float a_ = to_f32<T>(R[o]), b_ = to_f32<T>(R[o + (ldc >> 1)]);
            const float sg = 1.f / (1.f + __expf(-a_));
            C[o] = from_f32<TO>(v[e] * b_ * sg * (1.f + a_ * (1.f - sg)));
            C[o + (ldc >> 1)] = from_f32<TO>(v[e] * a_ * sg);
          } else {
            float x = v[e];
            if (EPI == FVQA_EPI_RESIDUAL) x += to_f32<T>(R[(size_t)m * ldc + n + e]);
            C[(size_t)m * ldc + n + e] = from_f32<TO>(x);
          }
        }
      }
    }
  }
}

// out = sum_s ws[s] (+R); rows >= m_split accumulate into tail (fp32)
template <typename T, typename TO, int EPI>
__global__ __launch_bounds__(256) void splitk_fixup(const float* __restrict__ ws, TO* __restrict__ C,
                                                    const T* __restrict__ R, float* __restrict__ tail, int M, int N,
                                                    int ldc, int m_split, int splits) {
  const size_t n4 = (size_t)M * (N / 4);
  const size_t plane = (size_t)M * N;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
    const int m = (int)(i / (N / 4));
    const int n = (int)(i % (N / 4)) * 4;
    float v[4];
    Vec4<float>::load(ws + (size_t)m * N + n, v);
    for (int s = 1; s < splits; ++s) {
      float u[4];
      Vec4<float>::load(ws + s * plane + (size_t)m * N + n, u);
      v[0] += u[0]; v[1] += u[1]; v[2] += u[2]; v[3] += u[3];
    }
    if (tail != nullptr && m >= m_split) {
      float* tp = tail + (size_t)(m - m_split) * N + n;
      float u[4];
      Vec4<float>::load(tp, u);
      u[0] += v[0]; u[1] += v[1]; u[2] += v[2]; u[3] += v[3];
      Vec4<float>::store(tp, u);
    } else {
      if (EPI == FVQA_EPI_SWIGLU_BWD) {           // v = dz; R = ab rows (a | b at +ldc/2); C = dab
        const size_t o = (size_t)m * ldc + n;
        float a_[4], b_[4], da[4], db[4];
        Vec4<T>::load(R + o, a_);
        Vec4<T>::load(R + o + (ldc >> 1), b_);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float sg = 1.f / (1.f + __expf(-a_[e]));
          da[e] = v[e] * b_[e] * sg * (1.f + a_[e] * (1.f - sg));
          db[e] = v[e] * a_[e] * sg;
        }
        Vec4<TO>::store(C + o, da);
        Vec4<TO>::store(C + o + (ldc >> 1), db);
        continue;
      }
      if (EPI == FVQA_EPI_RESIDUAL) {
        float r[4];
        Vec4<T>::load(R + (size_t)m * ldc + n, r);
        v[0] += r[0]; v[1] += r[1]; v[2] += r[2]; v[3] += r[3];
      }
      Vec4<TO>::store(C + (size_t)m * ldc + n, v);
    }
  }
}

template <typename T, typename TO, int EPI, int PIPE, int NT = 4>
int launch_256(const void* A, const void* B, void* C, const void* R, float* tail, float* ws, int M, int N, int K,
               int lda, int ldb, int ldc, int m_split, int splits, bool partial_only, hipStream_t st) {
  constexpr int TNW = 64 * NT;
  const int tm = (M + TM - 1) / TM, tn = (N + TNW - 1) / TNW;
  dim3 grid(tm * tn * splits), block(512);
  constexpr int RING = ((PIPE == 3 || PIPE == 6) ? 5 : PIPE == 2 ? 4 : PIPE == 0 ? NSTAGE : PIPE) * STAGE;
  constexpr int STG = 8 * 64 * 64 * 4;                  // epilogue staging: 8 waves x 64 rows x 64 floats
  constexpr int LDSB = RING > STG ? RING : STG;
  if (splits > 1 || partial_only) {
    auto k = gemm_nt_256<T, TO, EPI, true, PIPE, NT>;
    static bool attr_done = false;
    if (!attr_done) { (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, LDSB); attr_done = true; }
    {
      FvqaProbeScope ts(st, 2.0 * M * N * K, EPI | 16 | (sizeof(TO) == 4 ? 32 : 0) | (sizeof(T) == 4 ? 64 : 0));
      hipLaunchKernelGGL(k, grid, block, LDSB, st, (const T*)A, (const T*)B, (TO*)C, (const T*)R, tail, ws, M, N, K,
                         lda, ldb, ldc, m_split, tm, splits);
    }
    if (!partial_only) {
      size_t n4 = (size_t)M * (N / 4);
      int g = (int)((n4 + 255) / 256);
      if (g > 2048) g = 2048;
      hipLaunchKernelGGL((splitk_fixup<T, TO, EPI>), dim3(g), dim3(256), 0, st, (const float*)ws, (TO*)C,
                         (const T*)R, tail, M, N, ldc, m_split, splits);
    }
  } else {
    auto k = gemm_nt_256<T, TO, EPI, false, PIPE, NT>;
    static bool attr_done = false;
    if (!attr_done) { (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, LDSB); attr_done = true; }
    FvqaProbeScope ts(st, 2.0 * M * N * K, EPI | (sizeof(TO) == 4 ? 32 : 0) | (sizeof(T) == 4 ? 64 : 0));
    hipLaunchKernelGGL(k, grid, block, LDSB, st, (const T*)A, (const T*)B, (TO*)C, (const T*)R, tail, ws, M, N, K,
                       lda, ldb, ldc, m_split, tm, 1);
  }
  FVQA_CHECK_LAUNCH();
  return FVQA_OK;
}

}  // namespace

// how many K splits the 256-tile path uses for an (M, N, K) problem on a 256-CU part
extern "C" int fvqa_gemm_splits(int M, int N, int K, int dtype) {
  const int tiles = ((M + TM - 1) / TM) * ((N + TN - 1) / TN);
  const int ke = dtype == FVQA_BF16 ? 32 : 16;
  const int nk = K / ke;
  int s = 256 / (tiles > 0 ? tiles : 1);
  if (s < 1) s = 1;
  if (s > 8) s = 8;
  while (s > 1 && nk / s < 16) --s;       // keep >= 16 stages per split: the ring needs a run-up
  return s;
}

// Tail-round plan for outputs with MORE tiles than CUs: the whole N-tile columns that do not fit into the full
// rounds are computed by a second launch that splits K, so the launch pair takes `full + 1/sr` rounds instead of
// `full + 1` (M = 1024, W1|W3: 344 tiles -> 256 + 88 x 2 halves; three streams, M = 3072, W1|W3: 1032 tiles ->
// 1020 in 4 rounds + 12 x 8 eighths instead of 5 rounds).
struct TailPlan { int n1; int sr; int rounds1; };
static TailPlan tail_plan(int M, int N, int K, int dtype) {
  TailPlan p = {N, 1, 0};
  const int tm = (M + TM - 1) / TM, tn = (N + TN - 1) / TN;
  const int tiles = tm * tn;
  if (tiles <= 256 || (N % TN) != 0) return p;
  const int full = tiles / 256;
  if (tiles % 256 == 0) return p;
  const int c1 = full * 256 / tm;                       // N-tile columns that fit the full rounds
  const int c2 = tn - c1;
  if (c1 < 1 || c2 < 1) return p;
  int sr = 256 / (c2 * tm);
  if (sr > 8) sr = 8;
  const int nk = K / (dtype == FVQA_BF16 ? 32 : 16);
  while (sr > 1 && nk / sr < 16) --sr;                  // keep >= 16 stages per split: the ring needs a run-up
  if (sr < 2) return p;
  const int rounds1 = (tm * c1 + 255) / 256;
  if (rounds1 + 1.0 / sr + 0.1 > (tiles + 255) / 256 - 0.15) return p;     // must beat the plain launch
  p.n1 = c1 * TN;
  p.sr = sr;
  p.rounds1 = rounds1;
  return p;
}

extern "C" size_t fvqa_gemm_sk_workspace(void);
// bytes that serve every path fvqa_gemm_nt may take for the problem: the persistent kernel's flags + slabs, or
// the split-K planes of the older paths behind the 4 KiB flag block
extern "C" size_t fvqa_gemm_workspace(int M, int N, int K, int dtype) {
  const int s = fvqa_gemm_splits(M, N, K, dtype);
  size_t old_path = 0;
  if (s > 1) old_path = (size_t)s * M * N * sizeof(float);
  else {
    const TailPlan p = tail_plan(M, N, K, dtype);
    old_path = p.sr > 1 ? (size_t)p.sr * M * (N - p.n1) * sizeof(float) : 0;
  }
  old_path += 4096;
  const size_t sk = (M >= 192 && N >= 256) ? fvqa_gemm_sk_workspace() : 0;
  return old_path > sk ? old_path : sk;
}

// Tile width for an unsplit problem on 256 CUs: rounds x work per tile, 256-wide (with its tail-round
// plan where that applies: + 1/sr round + a fix-up pass) against 192-wide tiles at 3/4 of the work.
// M = 1024: W2^T (N = 11008) 172 x 1.0 -> 232 x 0.75; W1|W3 (N = 22016) 1.5 rounds + fix-up -> 2 x 0.75.
static int pick_nt(int M, int N, int K, int dtype, bool tail_plan_ok) {
  const int tm = (M + TM - 1) / TM;
  const int t256 = tm * ((N + 255) / 256), t192 = tm * ((N + 191) / 192);
  double c256 = (double)((t256 + 255) / 256);
  if (tail_plan_ok) {
    const TailPlan p = tail_plan(M, N, K, dtype);
    if (p.sr > 1) c256 = (double)p.rounds1 + 1.0 / p.sr + 0.1;
  }
  // measured (profiles/r01_gemm_ablation_tilewidth.log): at equal rounds the narrower tile buys nothing —
  // the loop is bound by LDS/DMA traffic per stage, which shrinks 9 %, not by MFMA work, which shrinks 25 %
  const double c192 = 0.95 * (double)((t192 + 255) / 256);
  return c192 < c256 - 0.2 ? 3 : 4;
}

// mode: PIPE of the kernel loop (0, 2, 3, 6); 61 / 63 = PIPE 6 with the tile width forced to 256 / 192
int fvqa_gemm_nt_256_impl(const void* A, const void* B, void* C, const void* R, float* tail, void* ws,
                          size_t ws_bytes, int M, int N, int K, int lda, int ldb, int ldc, int m_split, int dtype,
                          int out_dtype, int epilogue, int force_splits, int mode, hipStream_t st) {
  int force_nt = 0;
  if (mode == 61) { force_nt = 4; mode = 6; }
  if (mode == 63) { force_nt = 3; mode = 6; }
  if (epilogue == FVQA_EPI_SWIGLU_BWD) {       // elementwise epilogue on whole outputs
    const size_t ei = fvqa_dtype_size(dtype);
    if (force_splits == 0 && force_nt == 0 && ws != nullptr) {
      // tail-round plan: the columns past the full rounds run K-split and their fix-up pass applies SwiGLU'
      const TailPlan p = tail_plan(M, N, K, dtype);
      if (p.sr > 1 && ws_bytes >= (size_t)p.sr * M * (N - p.n1) * sizeof(float)) {
        int rc = fvqa_gemm_nt_256_impl(A, B, C, R, nullptr, ws, ws_bytes, M, p.n1, K, lda, ldb, ldc, m_split, dtype,
                                       out_dtype, epilogue, 1, 61, st);
        if (rc) return rc;
        return fvqa_gemm_nt_256_impl(A, (const char*)B + (size_t)p.n1 * ldb * ei, (char*)C + (size_t)p.n1 * ei,
                                     (const char*)R + (size_t)p.n1 * ei, nullptr, ws, ws_bytes, M, N - p.n1, K, lda,
                                     ldb, ldc, m_split, dtype, out_dtype, epilogue, p.sr, 61, st);
      }
    }
    const int nt = force_nt ? force_nt : pick_nt(M, N, K, dtype, false);
    const int sp = force_splits > 1 ? force_splits : 1;
    if (dtype == FVQA_BF16)
      return nt == 3 && sp == 1
                 ? launch_256<bf16_t, bf16_t, FVQA_EPI_SWIGLU_BWD, 6, 3>(A, B, C, R, nullptr, (float*)ws, M, N, K, lda,
                                                                        ldb, ldc, m_split, 1, false, st)
                 : launch_256<bf16_t, bf16_t, FVQA_EPI_SWIGLU_BWD, 6>(A, B, C, R, nullptr, (float*)ws, M, N, K, lda, ldb,
                                                                     ldc, m_split, sp, false, st);
    return nt == 3 && sp == 1
               ? launch_256<float, float, FVQA_EPI_SWIGLU_BWD, 6, 3>(A, B, C, R, nullptr, (float*)ws, M, N, K, lda, ldb,
                                                                    ldc, m_split, 1, false, st)
               : launch_256<float, float, FVQA_EPI_SWIGLU_BWD, 6>(A, B, C, R, nullptr, (float*)ws, M, N, K, lda, ldb, ldc,
                                                                 m_split, sp, false, st);
  }
  int splits = force_splits > 0 ? force_splits : fvqa_gemm_splits(M, N, K, dtype);
  const bool partial = epilogue == FVQA_EPI_PARTIAL;
  const bool plan_ok = !partial && force_splits == 0 && splits == 1 && tail == nullptr && ws != nullptr;
  int nt = 4;
  if (mode == 6 && !partial && splits == 1 && tail == nullptr)
    nt = force_nt ? force_nt : (force_splits == 0 ? pick_nt(M, N, K, dtype, plan_ok) : 4);
  if (plan_ok && nt == 4) {
    const TailPlan p = tail_plan(M, N, K, dtype);
    if (p.sr > 1 && ws_bytes >= (size_t)p.sr * M * (N - p.n1) * sizeof(float)) {
      const size_t eo = out_dtype == FVQA_F32 ? 4 : fvqa_dtype_size(dtype), ei = fvqa_dtype_size(dtype);
      int rc = fvqa_gemm_nt_256_impl(A, B, C, R, nullptr, ws, ws_bytes, M, p.n1, K, lda, ldb, ldc, m_split, dtype,
                                     out_dtype, epilogue, 1, mode, st);
      if (rc) return rc;
      return fvqa_gemm_nt_256_impl(A, (const char*)B + (size_t)p.n1 * ldb * ei, (char*)C + (size_t)p.n1 * eo,
                                   R ? (const char*)R + (size_t)p.n1 * ei : nullptr, nullptr, ws, ws_bytes, M,
                                   N - p.n1, K, lda, ldb, ldc, m_split, dtype, out_dtype, epilogue, p.sr, mode, st);
    }
  }
  if (partial) {
    if (!ws || ws_bytes < (size_t)splits * M * N * sizeof(float) || (N & 3)) return FVQA_EALIGN;
  } else if (splits > 1 && (!ws || ws_bytes < (size_t)splits * M * N * sizeof(float) || (N & 3))) {
    splits = 1;
  }
#define GO3(T, TO, P, NTV)                                                                                     \
  return epilogue == FVQA_EPI_RESIDUAL                                                                         \
             ? launch_256<T, TO, FVQA_EPI_RESIDUAL, P, NTV>(A, B, C, R, tail, (float*)ws, M, N, K, lda, ldb,    \
                                                            ldc, m_split, splits, false, st)                   \
             : launch_256<T, TO, FVQA_EPI_NONE, P, NTV>(A, B, C, R, tail, (float*)ws, M, N, K, lda, ldb, ldc,   \
                                                        m_split, splits, partial, st)
#define GO2(T, TO, P) GO3(T, TO, P, 4)
#define GO(T, TO)                                 \
  if (mode == 2) { GO2(T, TO, 2); }               \
  if (mode == 3) { GO2(T, TO, 3); }               \
  if (mode == 6 && nt == 3) { GO3(T, TO, 6, 3); } \
  if (mode == 6) { GO2(T, TO, 6); }               \
  GO2(T, TO, 0)
  if (dtype == FVQA_BF16) {
    if (out_dtype == FVQA_F32) { GO(bf16_t, float); }
    GO(bf16_t, bf16_t);
  }
  GO(float, float);
#undef GO
#undef GO2
#undef GO3
}
