// Decode-shape GEMM strip (M <= 16 rows, bf16): C[0..M, n0..n0+16) = A[M,K] · B[n0..n0+16, K]^T, computed by one
// 512-thread workgroup. HBM-bound on the weight stream, so the layout is chosen for bytes in flight, not for MFMA
// rate: the 8 waves split K eight ways (each keeps 8 k-steps = 16 KiB of weight + activation fragments in flight
// from global memory, no LDS staging) and the 8 partial 16x16 blocks meet in LDS (`part`, 10 KiB). The MFMA is fed the
// weight strip as its row operand, so a lane ends up with 4 consecutive columns of one row.
// Shared by the stand-alone kernel (gemm.hip: generation path, adapter rows) and by the side job that rides on the
// idle CUs of a persistent projection launch (gemm_sk.hip) — one arithmetic, bitwise-equal results.
// EPI: none / residual (C = acc + R) / FVQA_EPI_SKINNY_ACC (C += acc: gradient rows summed into an fp32 grad buffer).
#pragma once
#include "common.h"

constexpr int FVQA_EPI_SKINNY_ACC = 100;
constexpr int FVQA_SKINNY_LDS = 8 * 16 * 20 * 4;

template <typename TO, int EPI>
__device__ __forceinline__ void skinny_strip(const bf16_t* __restrict__ A, const bf16_t* __restrict__ B,
                                             TO* __restrict__ C, const bf16_t* __restrict__ R, int M, int N, int K,
                                             int lda, int ldb, int ldc, int n0, float (*part)[16][20]) {
  const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int li = lane & 15, g = lane >> 4;
  const int kw = K / 8;                                    // this wave's K range (K % 256 == 0)
  int bn = n0 + li; bn = bn < N ? bn : N - 1;
  int am = li < M ? li : M - 1;
  const bf16_t* bp = B + (size_t)bn * ldb + (size_t)w * kw + 8 * g;
  const bf16_t* ap = A + (size_t)am * lda + (size_t)w * kw + 8 * g;
  f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int k0 = 0; k0 < kw; k0 += 256) {
    uint4 bf[8], af[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int k = k0 + 32 * u;
      const bool in = k < kw;
      bf[u] = in ? *reinterpret_cast<const uint4*>(bp + k) : make_uint4(0, 0, 0, 0);
      af[u] = in ? *reinterpret_cast<const uint4*>(ap + k) : make_uint4(0, 0, 0, 0);
    }
#pragma unroll
    for (int u = 0; u < 8; ++u)                             // D[n = 4g+r][m = li]
      acc = FVQA_MFMA_H16_16x16x32(__builtin_bit_cast(h16x8_t, bf[u]),
                                                    __builtin_bit_cast(h16x8_t, af[u]), acc, 0, 0, 0);
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) part[w][li][4 * g + r] = acc[r];           // [wave][m][n]
  __syncthreads();
  if (threadIdx.x < 256) {
    const int m = threadIdx.x >> 4, n = threadIdx.x & 15;
    if (m < M && n0 + n < N) {
      float v = 0.f;
#pragma unroll
      for (int ww = 0; ww < 8; ++ww) v += part[ww][m][n];
      if (EPI == FVQA_EPI_RESIDUAL) v += to_f32<bf16_t>(R[(size_t)m * ldc + n0 + n]);
      if (EPI == FVQA_EPI_SKINNY_ACC) v += to_f32<TO>(C[(size_t)m * ldc + n0 + n]);
      C[(size_t)m * ldc + n0 + n] = from_f32<TO>(v);
    }
  }
}

// The same strips for a 256-thread workgroup (the side job of the 4-wave projection kernel, gemm4w_kernel.h): TWO adjacent
// strips (32 columns) per call, each wave taking two of the eight K ranges (w and w + 4) of both. The arithmetic of a strip is
// the 512-thread routine's, bit for bit: eight partial blocks, one per K range, each accumulated in k order, summed in range
// order. What differs is the memory pipeline: four independent accumulation chains per wave share the activation fragments,
// batches of 4 k-steps are double-buffered (the next batch's loads are issued before the current one's MFMAs) and the fp32
// rows a gradient rider adds to are requested before the first weight load — a workgroup with half the waves streams the
// weight rows at about the same rate (measured 14 us per strip before, tools/gemm4w_probe epi).
constexpr int FVQA_SKINNY2_LDS = 2 * 8 * 16 * 20 * 4;

template <typename TO, int EPI>
__device__ __forceinline__ void skinny_strip2_4w(const bf16_t* __restrict__ A, const bf16_t* __restrict__ B,
                                                 TO* __restrict__ C, int M, int N, int K, int lda, int ldb, int ldc, int n0,
                                                 float (*part)[8][16][20]) {
  const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int li = lane & 15, g = lane >> 4;
  const int kw = K / 8;
  const int om = threadIdx.x >> 4, on = threadIdx.x & 15;  // this thread's output (row, column within a strip) of the final sum
  float cin[2] = {0.f, 0.f};
  if (EPI == FVQA_EPI_SKINNY_ACC) {
#pragma unroll
    for (int s = 0; s < 2; ++s)
      if (om < M && n0 + 16 * s + on < N) cin[s] = to_f32<TO>(C[(size_t)om * ldc + n0 + 16 * s + on]);
  }
  int am = li < M ? li : M - 1;
  const bf16_t* bp[2][2]; const bf16_t* ap[2];
  f32x4 acc[2][2];
#pragma unroll
  for (int e = 0; e < 2; ++e) {
    ap[e] = A + (size_t)am * lda + (size_t)(w + 4 * e) * kw + 8 * g;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      int bn = n0 + 16 * s + li; bn = bn < N ? bn : N - 1;
      bp[s][e] = B + (size_t)bn * ldb + (size_t)(w + 4 * e) * kw + 8 * g;
      acc[s][e] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
  }
  uint4 b0[2][2][4], a0[2][4], b1[2][2][4], a1[2][4];
  auto ld = [&](uint4(&bb)[2][2][4], uint4(&aa)[2][4], int k0) {
#pragma unroll
    for (int e = 0; e < 2; ++e)
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int k = k0 + 32 * u;
        const bool in = k < kw;
        aa[e][u] = in ? *reinterpret_cast<const uint4*>(ap[e] + k) : make_uint4(0, 0, 0, 0);
#pragma unroll
        for (int s = 0; s < 2; ++s) bb[s][e][u] = in ? *reinterpret_cast<const uint4*>(bp[s][e] + k) : make_uint4(0, 0, 0, 0);
      }
  };
  auto mm = [&](const uint4(&bb)[2][2][4], const uint4(&aa)[2][4]) {
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int e = 0; e < 2; ++e)
#pragma unroll
        for (int u = 0; u < 4; ++u)                         // D[n = 4g+r][m = li]; every (strip, range) chain in k order
          acc[s][e] = FVQA_MFMA_H16_16x16x32(__builtin_bit_cast(h16x8_t, bb[s][e][u]),
                                                              __builtin_bit_cast(h16x8_t, aa[e][u]), acc[s][e], 0, 0, 0);
  };
  ld(b0, a0, 0);
  for (int k0 = 0; k0 < kw; k0 += 256) {
    ld(b1, a1, k0 + 128);
    mm(b0, a0);
    ld(b0, a0, k0 + 256);
    mm(b1, a1);
  }
#pragma unroll
  for (int s = 0; s < 2; ++s)
#pragma unroll
    for (int e = 0; e < 2; ++e)
#pragma unroll
      for (int r = 0; r < 4; ++r) part[s][w + 4 * e][li][4 * g + r] = acc[s][e][r];      // [strip][K range][m][n]
  __syncthreads();
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    const int n = n0 + 16 * s + on;
    if (om < M && n < N) {
      float v = 0.f;
#pragma unroll
      for (int ww = 0; ww < 8; ++ww) v += part[s][ww][om][on];
      if (EPI == FVQA_EPI_SKINNY_ACC) v += cin[s];
      C[(size_t)om * ldc + n] = from_f32<TO>(v);
    }
  }
}

// ---- the same two strips with the operands streamed by LDS-DMA (round 5) ------------------------------------------------------------
// skinny_strip2_4w keeps its weight fragments in flight in registers — and hipcc drains the queue (`s_waitcnt vmcnt(0)`) inside
// its double-buffered loop, so a light workgroup streams ~17.5 GB/s beside the busy CUs of a projection launch: the adapter K/V
// rider held the W1|W3 launch 18 us past its tiles (1.28 MB per light workgroup in the 75 us its second round lasts) and forced
// the dH W2^T launch onto 13-block tiles (tools/gemm4w_widths.py with GW_NO_RIDER=1: 149.8 against 167.7 us, 82.0 against 91.6).
// Here every wave streams the rows of ITS two K ranges — weights AND the <= 16 activation rows — through a private ring in the
// (idle) LDS of the workgroup: per stage (2 strips + 1 activation block) x 2 ranges x 16 rows x 128 B = 12 KiB as twelve 1-KiB
// LDS-DMA pieces (8 rows x 128 B = whole cache lines per row; lane-linear image, chunk c of row r at c ^ (r & 7): conflict-free
// ds_read_b128 fragments, as the GEMM rings), FVQA_SKINNY_DMA_STAGES stages deep: two stages (24 KiB per wave, 96 KiB per
// workgroup) in flight while one is consumed, no barrier inside the K loop (a wave reads only what it loaded itself; its own
// counted vmcnt covers its pieces). No load of the loop has a register destination: every vector-memory operation is an LDS-DMA
// builtin, counted by hand, and there is nothing for the compiler to copy, spill or drain (a first form with inline-asm
// activation loads let hipcc move their destination registers while the loads were in flight: garbage on three waves of four).
// ARITHMETIC: unchanged — (strip, K range) chains accumulated in k order, the eight ranges summed in range order: bit for bit the
// results of skinny_strip2_4w and of the stand-alone skinny_strip. Needs (K / 8) % 64 == 0 (the caller falls back otherwise).
// LDS: 4 waves x 3 stages x 12 KiB = 144 KiB; the eight partial blocks (`part`, 20 KiB) alias the ring once every wave is done
// with it (one barrier).
constexpr int FVQA_SKINNY_DMA_STAGES = 3;
constexpr int FVQA_SKINNY_DMA_WSTAGE = 12 * 1024;                                        // bytes per wave per stage
constexpr int FVQA_SKINNY_DMA_LDS = 4 * FVQA_SKINNY_DMA_STAGES * FVQA_SKINNY_DMA_WSTAGE; // 144 KiB per workgroup
static_assert(FVQA_SKINNY_DMA_LDS >= FVQA_SKINNY2_LDS, "the partial blocks alias the ring");

#ifndef FVQA_SKINNY_DMA_AUX
#define FVQA_SKINNY_DMA_AUX 0
#endif

template <typename TO, int EPI>
__device__ __forceinline__ void skinny_strip2_dma_4w(const bf16_t* __restrict__ A, const bf16_t* __restrict__ B,
                                                     TO* __restrict__ C, int M, int N, int K, int lda, int ldb, int ldc, int n0,
                                                     char* ring) {
  typedef __attribute__((address_space(1))) const void gptr_t;
  typedef __attribute__((address_space(3))) void lptr_t;
  constexpr int S = FVQA_SKINNY_DMA_STAGES, GROUP = 12;      // LDS-DMA pieces per stage and wave
  float(*part)[8][16][20] = reinterpret_cast<float(*)[8][16][20]>(ring);
  const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int li = lane & 15, g = lane >> 4;
  const int kw = K / 8;
  const int nst = kw / 64;                                   // stages: 64 k elements (two k-steps) of each of this wave's ranges
  const int om = threadIdx.x >> 4, on = threadIdx.x & 15;
  float cin[2] = {0.f, 0.f};
  if (EPI == FVQA_EPI_SKINNY_ACC) {
#pragma unroll
    for (int s = 0; s < 2; ++s)
      if (om < M && n0 + 16 * s + on < N) cin[s] = to_f32<TO>(C[(size_t)om * ldc + n0 + 16 * s + on]);
  }
  const int lr = lane >> 3, lc = (lane & 7) ^ lr;            // DMA lane: row within an 8-row piece, SOURCE chunk of its slot
  // source of block b (0, 1: the two weight strips; 2: the activation rows), range e, half h (rows 8h .. 8h + 7), stage 0
  const bf16_t* src[3][2][2];
  f32x4 acc[2][2];
#pragma unroll
  for (int e = 0; e < 2; ++e)
#pragma unroll
    for (int h = 0; h < 2; ++h) {
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        int bn = n0 + 16 * s + 8 * h + lr; bn = bn < N ? bn : N - 1;
        src[s][e][h] = B + (size_t)bn * ldb + (size_t)(w + 4 * e) * kw + 8 * lc;
      }
      int am = 8 * h + lr; am = am < M ? am : M - 1;
      src[2][e][h] = A + (size_t)am * lda + (size_t)(w + 4 * e) * kw + 8 * lc;
    }
#pragma unroll
  for (int s = 0; s < 2; ++s)
#pragma unroll
    for (int e = 0; e < 2; ++e) acc[s][e] = f32x4{0.f, 0.f, 0.f, 0.f};
  char* const myring = ring + w * (S * FVQA_SKINNY_DMA_WSTAGE);
  // fragment read offset inside a (block, range) pair of 1-KiB pieces: row li -> piece li >> 3, row-in-piece li & 7
  const int r8 = li & 7;
  const int rd0 = (li >> 3) * 1024 + r8 * 128 + (((0 + g) ^ r8) << 4);      // k-step 0: chunk g
  const int rd1 = (li >> 3) * 1024 + r8 * 128 + (((4 + g) ^ r8) << 4);      // k-step 1: chunk 4 + g
  auto issue = [&](int t) {                                  // one group: GROUP LDS-DMA pieces into slot t % S
    char* dst = myring + (t % S) * FVQA_SKINNY_DMA_WSTAGE;
#pragma unroll
    for (int b = 0; b < 3; ++b)
#pragma unroll
      for (int e = 0; e < 2; ++e)
#pragma unroll
        for (int h = 0; h < 2; ++h)
          __builtin_amdgcn_global_load_lds((gptr_t*)(src[b][e][h] + (size_t)t * 64), (lptr_t*)(dst + ((b * 2 + e) * 2 + h) * 1024), 16, 0,
                                           FVQA_SKINNY_DMA_AUX);
  };
  // wait until the group of stage t has landed: `later` groups were issued after it (GROUP operations each, retired in order)
  auto landed = [&](int later) {
    if (later >= 2) asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
    else if (later == 1) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
  };
  static_assert(GROUP == 12 && S == 3, "the counted waits above are written for two groups of 12 in flight");
  auto compute = [&](int t) {
    const char* slot = myring + (t % S) * FVQA_SKINNY_DMA_WSTAGE;
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const char* ablk = slot + (2 * 2 + e) * 2048;
      const uint4 x0 = *reinterpret_cast<const uint4*>(ablk + rd0), x1 = *reinterpret_cast<const uint4*>(ablk + rd1);
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const char* blk = slot + (s * 2 + e) * 2048;
        const uint4 f0 = *reinterpret_cast<const uint4*>(blk + rd0), f1 = *reinterpret_cast<const uint4*>(blk + rd1);
        acc[s][e] = FVQA_MFMA_H16_16x16x32(__builtin_bit_cast(h16x8_t, f0), __builtin_bit_cast(h16x8_t, x0), acc[s][e], 0, 0, 0);
        acc[s][e] = FVQA_MFMA_H16_16x16x32(__builtin_bit_cast(h16x8_t, f1), __builtin_bit_cast(h16x8_t, x1), acc[s][e], 0, 0, 0);
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // the slot's reads have returned before a later group refills it
  };
  // (everything the wave has in flight — the tile's stores, the rows above — leaves the count before the first group enters it)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (0 < nst) issue(0);
  if (1 < nst) issue(1);
  for (int t = 0; t < nst; ++t) {
    if (t + 2 < nst) issue(t + 2);                           // into the slot stage t - 1 has just left
    const int later = nst - 1 - t;
    landed(later < 2 ? later : 2);
    compute(t);
  }
  __syncthreads();                                           // every wave is done with its ring: `part` may take its place
#pragma unroll
  for (int s = 0; s < 2; ++s)
#pragma unroll
    for (int e = 0; e < 2; ++e)
#pragma unroll
      for (int r = 0; r < 4; ++r) part[s][w + 4 * e][li][4 * g + r] = acc[s][e][r];      // [strip][K range][m][n]
  __syncthreads();
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    const int n = n0 + 16 * s + on;
    if (om < M && n < N) {
      float v = 0.f;
#pragma unroll
      for (int ww = 0; ww < 8; ++ww) v += part[s][ww][om][on];
      if (EPI == FVQA_EPI_SKINNY_ACC) v += cin[s];
      C[(size_t)om * ldc + n] = from_f32<TO>(v);
    }
  }
}
