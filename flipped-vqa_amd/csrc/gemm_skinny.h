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
  typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
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
      acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, bf[u]),
                                                    __builtin_bit_cast(bf16x8_t, af[u]), acc, 0, 0, 0);
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
