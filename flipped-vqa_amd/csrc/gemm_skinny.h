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
