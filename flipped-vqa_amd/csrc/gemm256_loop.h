// Main loop of the 256-row projection GEMM for gfx950 (used by the persistent kernel in gemm_sk.hip).
//
// One call accumulates acc += A[m0.., k-range] x B[n0.., k-range]^T for one 256 x (64*NT) output tile over
// `nw` WIDE stages (a wide stage = 128 bytes of K per row = 64 bf16 / 32 fp32 = two MFMA k-steps).
// Structure (512 threads = 8 waves as 2(M) x 4(N), 128 x 16*NT outputs per wave, MFMA 16x16x32 bf16 /
// 16x16x4 f32 in NT form with the WEIGHT fragment as the row operand, so a lane holds 4 consecutive output
// columns of one row):
//   * asymmetric LDS-DMA rings: activations A 2 x 32 KiB, weights B 3 x (8*NT) KiB; the DMA is role-split
//     (waves 0-3 move A one stage ahead, waves 4-7 move B two stages ahead: vmcnt retires in order per
//     wave, so the deep HBM stream of the once-read weight panel gets its own waves);
//   * K rows of 128 bytes so that every global_load_lds_dwordx4 fetches whole cache lines; the LDS image is
//     lane-linear (DMA constraint), XOR-swizzled on the SOURCE address and on the ds_read_b128 address;
//   * one raw s_barrier per wide stage, counted s_waitcnt vmcnt (never 0 in steady state for the B waves);
//   * waves 4-7 run half a stage behind waves 0-3 (their k-step-1 MFMAs of rows 0-63 are deferred past the
//     next barrier), so the matrix pipe does not drain at every barrier;
//   * LDS fragment reads are inline asm with counted lgkmcnt waits: hipcc would otherwise make every
//     compiler-visible LDS load wait for ALL in-flight LDS-DMA (vmcnt(0)) and drain the ring each stage.
// On return every DMA of the call has landed and been consumed by THIS wave's reads; the caller must
// __syncthreads() before reusing the ring memory for anything else.
#pragma once
#include "common.h"

namespace fvqa_ring {
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int TM = 256;

template <typename T> struct Mma;
template <> struct Mma<bf16_t> {
  static constexpr int KE = 32;          // elements per MFMA k-step (64 bytes of a row)
  static __device__ __forceinline__ void run(const u32x4& a, const u32x4& b, f32x4& acc) {
    acc = FVQA_MFMA_H16_16x16x32(__builtin_bit_cast(h16x8_t, b),
                                                  __builtin_bit_cast(h16x8_t, a), acc, 0, 0, 0);
  }
};
template <> struct Mma<float> {
  static constexpr int KE = 16;
  static __device__ __forceinline__ void run(const u32x4& a, const u32x4& b, f32x4& acc) {
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(b.x), __uint_as_float(a.x), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(b.y), __uint_as_float(a.y), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(b.z), __uint_as_float(a.z), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(b.w), __uint_as_float(a.w), acc, 0, 0, 0);
  }
};

// ring bytes: A 2 x 32 KiB + B 3 x 32 KiB
constexpr int RING_BYTES = 5 * 32768;

// A, B: row-major with K contiguous; kel0 = first K element of the range; nw = wide stages in the range.
// rowxor (a multiple of 16, < 128): the tile's rows are held permuted — register block i of a wave's 128 rows is
// tile row block i ^ (rowxor / 16) — which lets each split-K partner keep the rows it will reduce in acc[0..].
template <typename T, int NT>
__device__ __forceinline__ void ring_loop(f32x4 (&acc)[8][NT], char* smem, const T* __restrict__ A,
                                          const T* __restrict__ B, int M, int N, int lda, int ldb, int m0, int n0,
                                          size_t kel0, int nw, int w, int lane, int rowxor = 0) {
  constexpr int KE = Mma<T>::KE;
  constexpr int CH = 16 / (int)sizeof(T);
  constexpr int WN = 16 * NT;              // output columns per wave
  const int wr = w >> 2, wc = w & 3;
  const int frow = lane & 15;
  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
#define FVQA_DSR(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "i"(off))
    constexpr int WROW = 128, WOP = TM * WROW;            // A: 32 KiB per stage
    constexpr int WOPB = 4 * WN * WROW;                   // B: 32 KiB (NT 4) / 24 KiB (NT 3) per stage
    constexpr int NPB = 2 * NT;                           // 1-KiB pieces per B wave per stage
    const bool bwave = w >= 4;                            // wave-uniform DMA role
    const int wq = w & 3;
    const T* wsrc[8];
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      // pieces of 8 rows x 128 B: 32 per A stage (8 per wave), 8*NT per B stage (NPB per wave)
      const int piece = bwave ? wq * NPB + (t < NPB ? t : NPB - 1) : wq * 8 + t;
      const int row = piece * 8 + (lane >> 3);
      const int c = (lane & 7) ^ (row & 7);
      int ga = m0 + (row ^ rowxor); ga = ga < M ? ga : M - 1;   // (rowxor: LDS row r holds tile row r ^ rowxor)
      int gb = n0 + row; gb = gb < N ? gb : N - 1;
      wsrc[t] = bwave ? B + (size_t)gb * ldb + kel0 + c * CH
                      : A + (size_t)ga * lda + kel0 + c * CH;
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
#define FVQA_WROW(i, n)                                                                \
  asm volatile("s_waitcnt lgkmcnt(" #n ")" : "+v"(a[i]));                              \
  Mma<T>::run(a[i], b[0], acc[i][0]); Mma<T>::run(a[i], b[1], acc[i][1]);        \
  Mma<T>::run(a[i], b[2], acc[i][2]);                                               \
  if constexpr (NT == 4) Mma<T>::run(a[i], b[3], acc[i][NT - 1]);                   \
  __builtin_amdgcn_sched_barrier(0);
  // the DMA issues of the next stage ride one per MFMA row of k-step 0 (an LDS-DMA issue costs ~100 cycles of
  // the wave's issue stream; behind an MFMA row it hides)
#define FVQA_WROW_DMA(i, n, q)                                                         \
  FVQA_WROW(i, n)                                                                      \
  if (more && (!bwave || q < NPB)) issue_piece(nu, nslot, q);                          \
  __builtin_amdgcn_sched_barrier(0);
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
  Mma<T>::run(hl[i], hb[0], acc[i][0]); Mma<T>::run(hl[i], hb[1], acc[i][1]);    \
  Mma<T>::run(hl[i], hb[2], acc[i][2]);                                             \
  if constexpr (NT == 4) Mma<T>::run(hl[i], hb[3], acc[i][NT - 1]);
#define FVQA_UROW(i)                                                                   \
  Mma<T>::run(au[i - 4], hb[0], acc[i][0]); Mma<T>::run(au[i - 4], hb[1], acc[i][1]); \
  Mma<T>::run(au[i - 4], hb[2], acc[i][2]);                                              \
  if constexpr (NT == 4) Mma<T>::run(au[i - 4], hb[3], acc[i][NT - 1]);
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
#undef FVQA_DSR
}

}  // namespace fvqa_ring
