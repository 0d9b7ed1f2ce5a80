// Frozen-weight projection GEMM for gfx950:  C[M,N] = A[M,K] · B[N,K]^T (+R).
//
// Replaces every F.linear on the training path (reference llama/model.py:89,99-100,127-128,142,
// 348) and, through host-side transposed weight copies, every autograd dX = dY·W of them.
// Both operands are K-contiguous, so a 16-byte LDS chunk is an MFMA fragment for A and for B.
//
// Kernel family `gemm_nt_128`: 128x128 output tile per 256-thread workgroup (4 waves as 2x2,
// 64x64 per wave = 4x4 MFMA 16x16 tiles), K staged in 128-byte rows (64 bf16 / 32 fp32) through
// a double-buffered 64 KiB LDS image filled by direct global->LDS DMA (global_load_lds_dwordx4).
// The LDS image is lane-linear; the XOR swizzle that makes the ds_read_b128 fragment reads
// bank-conflict-free is applied on the per-lane SOURCE address and again on the read.
//   bf16 : v_mfma_f32_16x16x32_bf16 (dense peak ~2.5 PF)
//   fp32 : v_mfma_f32_16x16x4_f32  (exact fp32 fma chain; the validation build)
// Workgroup ids are remapped so that each XCD (private L2) owns a contiguous run of tiles that
// walk M fastest: the 8..24 row-tiles sharing one weight panel hit that panel in one L2.
#include "common.h"
#include "gemm_skinny.h"

namespace {

constexpr int BM = 128;
constexpr int BN = 128;
constexpr int ROWB = 128;               // bytes of K per LDS row
constexpr int TILE_BYTES = BM * ROWB;   // 16 KiB per operand per buffer
constexpr int GEMM_LDS = 4 * TILE_BYTES;

template <typename T> struct Mma;
template <> struct Mma<bf16_t> {
  static constexpr int KELEMS = 64;
  static __device__ __forceinline__ void run(const uint4& a, const uint4& b, f32x4& acc) {
    acc = FVQA_MFMA_H16_16x16x32(__builtin_bit_cast(h16x8_t, a),
                                                  __builtin_bit_cast(h16x8_t, b), acc, 0, 0, 0);
  }
};
template <> struct Mma<float> {
  static constexpr int KELEMS = 32;
  // a 16-byte chunk holds k = 4g..4g+3 for lane group g = lane>>4: MFMA t consumes element t of
  // both operands, i.e. k-index 4g+t on lane group g — the same k on the A and the B side.
  static __device__ __forceinline__ void run(const uint4& a, const uint4& b, f32x4& acc) {
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.x), __uint_as_float(b.x), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.y), __uint_as_float(b.y), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.z), __uint_as_float(b.z), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.w), __uint_as_float(b.w), acc, 0, 0, 0);
  }
};

// bijective XCD-aware remap (blocks b and b+8 share an XCD under round-robin dispatch)
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
  const int q = nwg >> 3, r = nwg & 7, x = bid & 7;
  const int base = (x < r) ? x * (q + 1) : r * (q + 1) + (x - r) * q;
  return base + (bid >> 3);
}

template <typename T, typename TO, bool GLDS, int EPI>
__global__ __launch_bounds__(256) void gemm_nt_128(const T* __restrict__ A, const T* __restrict__ B,
                                                   TO* __restrict__ C, const T* __restrict__ R,
                                                   float* __restrict__ tail, int M, int N, int K, int lda,
                                                   int ldb, int ldc, int m_split, int tiles_m) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int KE = Mma<T>::KELEMS;
  constexpr int CH = 16 / (int)sizeof(T);   // elements per 16-byte chunk

  const int tile = xcd_remap(blockIdx.x, gridDim.x);
  const int m0 = (tile % tiles_m) * BM;
  const int n0 = (tile / tiles_m) * BN;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int wr = w >> 1, wc = w & 1;

  // ---- staging geometry: wave w moves pieces w*4 .. w*4+3 (1 KiB = 8 rows x 128 B each)
  const T* srcA[4];
  const T* srcB[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int piece = w * 4 + t;
    const int row = piece * 8 + (lane >> 3);
    const int c = (lane & 7) ^ (row & 7);            // source chunk for LDS chunk position lane&7
    int ga = m0 + row; ga = ga < M ? ga : M - 1;
    int gb = n0 + row; gb = gb < N ? gb : N - 1;
    srcA[t] = A + (size_t)ga * lda + c * CH;
    srcB[t] = B + (size_t)gb * ldb + c * CH;
  }
  auto stage_glds = [&](int buf, int kt) {
    char* dA = smem + buf * 2 * TILE_BYTES;
    char* dB = dA + TILE_BYTES;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int piece = w * 4 + t;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(srcA[t] + (size_t)kt * KE),
                                       (__attribute__((address_space(3))) void*)(dA + piece * 1024), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(srcB[t] + (size_t)kt * KE),
                                       (__attribute__((address_space(3))) void*)(dB + piece * 1024), 16, 0, 0);
    }
  };
  uint4 ra[4], rb[4];
  auto stage_load = [&](int kt) {
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      ra[t] = *reinterpret_cast<const uint4*>(srcA[t] + (size_t)kt * KE);
      rb[t] = *reinterpret_cast<const uint4*>(srcB[t] + (size_t)kt * KE);
    }
  };
  auto stage_write = [&](int buf) {
    char* dA = smem + buf * 2 * TILE_BYTES;
    char* dB = dA + TILE_BYTES;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int piece = w * 4 + t;
      *reinterpret_cast<uint4*>(dA + piece * 1024 + lane * 16) = ra[t];
      *reinterpret_cast<uint4*>(dB + piece * 1024 + lane * 16) = rb[t];
    }
  };

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // fragment read offsets (bytes inside an operand tile); row&7 == lane&7 for every fragment row
  const int frow = lane & 15;
  const int fsw = lane & 7;
  const int fkc = lane >> 4;
  const int offA = (wr * 64 + frow) * ROWB;
  const int offB = (wc * 64 + frow) * ROWB;

  const int nkt = K / KE;
  if (GLDS) {
    stage_glds(0, 0);
  } else {
    stage_load(0);
    stage_write(0);
  }
  __syncthreads();   // (drains the LDS DMA: hipcc emits vmcnt(0) ahead of the barrier)

  for (int kt = 0; kt < nkt; ++kt) {
    const int cur = kt & 1;
    const bool more = kt + 1 < nkt;
    if (more) {
      if (GLDS) stage_glds(cur ^ 1, kt + 1);
      else stage_load(kt + 1);
    }
    const char* sA = smem + cur * 2 * TILE_BYTES;
    const char* sB = sA + TILE_BYTES;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int coff = ((ks * 4 + fkc) ^ fsw) << 4;
      uint4 a[4], b[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) a[i] = *reinterpret_cast<const uint4*>(sA + offA + i * 16 * ROWB + coff);
#pragma unroll
      for (int j = 0; j < 4; ++j) b[j] = *reinterpret_cast<const uint4*>(sB + offB + j * 16 * ROWB + coff);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) Mma<T>::run(a[i], b[j], acc[i][j]);
    }
    if (!GLDS && more) stage_write(cur ^ 1);
    __syncthreads();
  }

  // ---- epilogue: C/D map of the 16x16 MFMA: col = lane&15, row = (lane>>4)*4 + reg
  const int ccol = lane & 15;
  const int crow = (lane >> 4) * 4;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
#pragma unroll
    for (int rg = 0; rg < 4; ++rg) {
      const int m = m0 + wr * 64 + i * 16 + crow + rg;
      if (m >= M) continue;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int n = n0 + wc * 64 + j * 16 + ccol;
        if (n >= N) continue;
        float v = acc[i][j][rg];
        if (tail != nullptr && m >= m_split) {
          tail[(size_t)(m - m_split) * N + n] += v;   // accumulates (grad buffer)
        } else {
          if (EPI == FVQA_EPI_RESIDUAL) v += to_f32<T>(R[(size_t)m * ldc + n]);
          C[(size_t)m * ldc + n] = from_f32<TO>(v);
        }
      }
    }
  }
}

template <typename T, typename TO, bool GLDS>
int launch_128(const void* A, const void* B, void* C, const void* R, float* tail, int M, int N, int K, int lda,
               int ldb, int ldc, int m_split, int epi, hipStream_t st) {
  const int tm = (M + BM - 1) / BM, tn = (N + BN - 1) / BN;
  dim3 grid(tm * tn), block(256);
  if (epi == FVQA_EPI_RESIDUAL) {
    auto k = gemm_nt_128<T, TO, GLDS, FVQA_EPI_RESIDUAL>;
    hipLaunchKernelGGL(k, grid, block, GEMM_LDS, st, (const T*)A, (const T*)B, (TO*)C, (const T*)R, tail, M, N,
                       K, lda, ldb, ldc, m_split, tm);
  } else {
    auto k = gemm_nt_128<T, TO, GLDS, FVQA_EPI_NONE>;
    hipLaunchKernelGGL(k, grid, block, GEMM_LDS, st, (const T*)A, (const T*)B, (TO*)C, (const T*)R, tail, M, N,
                       K, lda, ldb, ldc, m_split, tm);
  }
  FVQA_CHECK_LAUNCH();
  return FVQA_OK;
}

// ---- decode-shape GEMM (M <= 16 rows: one new token per sequence; adapter rows): one strip of 16 output columns per
// workgroup (gemm_skinny.h)
template <typename TO, int EPI>
__global__ __launch_bounds__(512) void gemm_nt_skinny(const bf16_t* __restrict__ A, const bf16_t* __restrict__ B,
                                                      TO* __restrict__ C, const bf16_t* __restrict__ R, int M, int N,
                                                      int K, int lda, int ldb, int ldc) {
  __shared__ float part[8][16][20];
  skinny_strip<TO, EPI>(A, B, C, R, M, N, K, lda, ldb, ldc, blockIdx.x * 16, part);
}

template <typename TO>
int launch_skinny(const void* A, const void* B, void* C, const void* R, int M, int N, int K, int lda, int ldb, int ldc,
                  int epi, hipStream_t st) {
  dim3 grid((N + 15) / 16), block(512);
  if (epi == FVQA_EPI_RESIDUAL)
    hipLaunchKernelGGL((gemm_nt_skinny<TO, FVQA_EPI_RESIDUAL>), grid, block, 0, st, (const bf16_t*)A,
                       (const bf16_t*)B, (TO*)C, (const bf16_t*)R, M, N, K, lda, ldb, ldc);
  else if (epi == FVQA_EPI_SKINNY_ACC)
    hipLaunchKernelGGL((gemm_nt_skinny<TO, FVQA_EPI_SKINNY_ACC>), grid, block, 0, st, (const bf16_t*)A,
                       (const bf16_t*)B, (TO*)C, (const bf16_t*)R, M, N, K, lda, ldb, ldc);
  else
    hipLaunchKernelGGL((gemm_nt_skinny<TO, FVQA_EPI_NONE>), grid, block, 0, st, (const bf16_t*)A, (const bf16_t*)B,
                       (TO*)C, (const bf16_t*)R, M, N, K, lda, ldb, ldc);
  FVQA_CHECK_LAUNCH();
  return FVQA_OK;
}

}  // namespace

extern "C" size_t fvqa_gemm_sk_workspace(void);
int fvqa_gemm_sk_impl(const void* A, const void* B, void* C, const void* R, void* ws, size_t ws_bytes, int M, int N,
                      int K, int lda, int ldb, int ldc, int dtype, int out_dtype, int epilogue, hipStream_t st,
                      const fvqa_sk_rider* rider, int* rode, void* C2 = nullptr, const fvqa_sk_rope* rope = nullptr);

// Problems the persistent family (gemm_sk.hip / gemm4w.hip) takes under variant 0: whole 256-row tiles — or ONE ragged row tile over a
// weight matrix of a 7B-class layer (>= 16 M elements; round 5: the tail rows — the last layer's post-attention projections, the LM
// head and their dX on the few dozen to few hundred rows a head reads). Such a launch is a weight stream: every tile cut along K so
// that 128-256 workgroups pull the matrix once (WO 4096 x 4096 at 33 rows: 128 x 128 kernel 67 us on 32 workgroups; LM head dX 590 us)
static inline bool persistent_shape(int M, int N, int K) {
  return N >= 256 && (M >= 192 || (M > 16 && (size_t)N * (size_t)K >= ((size_t)1 << 24)));
}

extern "C" size_t fvqa_gemm_workspace(int M, int N, int K, int dtype) {
  (void)dtype;
  return persistent_shape(M, N, K) ? fvqa_gemm_sk_workspace() : 0;
}

// variant: 0 = auto — the persistent 256-row kernel (gemm_sk.hip) for large problems it can store in whole 16-byte
// chunks, the decode-shape kernel for M <= 16 (bf16), else the 128x128 kernel; 1 = 128x128 register-staged;
// 2 = 128x128 LDS-DMA; 12 = the decode-shape (M <= 16) kernel; 13 = the persistent kernel (tests, tuning).
extern "C" int fvqa_gemm_nt(const void* A, const void* B, void* C, const void* R, float* tail, int M, int N,
                            int K, int lda, int ldb, int ldc, int m_split, int dtype, int out_dtype, int epilogue,
                            int variant, void* workspace, size_t workspace_bytes, void* stream) {
  if (!A || !B || (!C && !(tail && m_split == 0))) return FVQA_EINVAL;
  if (!fvqa_dtype_ok(dtype) || !fvqa_dtype_ok(out_dtype)) return FVQA_EINVAL;
  const bool swb = epilogue == FVQA_EPI_SWIGLU_BWD || epilogue == FVQA_EPI_SWIGLU_BWD_ST;
  if (epilogue != FVQA_EPI_NONE && epilogue != FVQA_EPI_RESIDUAL && !swb) return FVQA_EINVAL;
  if ((epilogue == FVQA_EPI_RESIDUAL || swb) && (!R || out_dtype != dtype)) return FVQA_EINVAL;
  if (swb && (tail || ldc != 2 * N)) return FVQA_EINVAL;
  if (out_dtype != dtype && out_dtype != FVQA_F32) return FVQA_EINVAL;
  if (variant != 0 && variant != 1 && variant != 2 && variant != 12 && variant != 13) return FVQA_EINVAL;
  if (M <= 0 || N <= 0 || K <= 0) return FVQA_ESHAPE;
  const int ke = dtype == FVQA_H16 ? 64 : 32;
  if (K % ke) return FVQA_ESHAPE;
  const size_t es = fvqa_dtype_size(dtype);
  if (((uintptr_t)A & 15) || ((uintptr_t)B & 15) || ((size_t)lda * es & 15) || ((size_t)ldb * es & 15))
    return FVQA_EALIGN;
  if (lda < K || ldb < K || ldc < N) return FVQA_ESHAPE;
  hipStream_t st = (hipStream_t)stream;
  const bool sk_ok = tail == nullptr && (N & 7) == 0 && (ldc & 7) == 0 && C &&
                     (((uintptr_t)C | (uintptr_t)R) & 15) == 0 && workspace != nullptr &&
                     ((uintptr_t)workspace & 255) == 0 && workspace_bytes >= fvqa_gemm_sk_workspace();
  if (variant == 13 && !sk_ok) return FVQA_EALIGN;
  if (variant == 13 || (variant == 0 && sk_ok && (persistent_shape(M, N, K) || swb)))
    return fvqa_gemm_sk_impl(A, B, C, R, workspace, workspace_bytes, M, N, K, lda, ldb, ldc, dtype, out_dtype, epilogue, st,
                             nullptr, nullptr);
  if (swb) return FVQA_EALIGN;                                // that epilogue lives in the persistent kernel only
  // every row goes to the fp32 tail (m_split == 0): the decode-shape kernel accumulates straight into it
  if (dtype == FVQA_H16 && M <= 16 && (K % 256) == 0 && tail != nullptr && m_split == 0 && epilogue == FVQA_EPI_NONE &&
      (variant == 0 || variant == 12))
    return launch_skinny<float>(A, B, tail, nullptr, M, N, K, lda, ldb, N, FVQA_EPI_SKINNY_ACC, st);
  const bool skinny_ok = dtype == FVQA_H16 && M <= 16 && (K % 256) == 0 && tail == nullptr &&
                         (epilogue == FVQA_EPI_NONE || epilogue == FVQA_EPI_RESIDUAL);
  if (variant == 12 && !skinny_ok) return FVQA_ESHAPE;
  if (skinny_ok && (variant == 0 || variant == 12)) {       // decode shape: stream the weights once at HBM speed
    return out_dtype == FVQA_F32 ? launch_skinny<float>(A, B, C, R, M, N, K, lda, ldb, ldc, epilogue, st)
                                 : launch_skinny<bf16_t>(A, B, C, R, M, N, K, lda, ldb, ldc, epilogue, st);
  }
  const bool glds = variant != 1;
  if (dtype == FVQA_H16) {
    if (out_dtype == FVQA_F32)
      return glds ? launch_128<bf16_t, float, true>(A, B, C, R, tail, M, N, K, lda, ldb, ldc, m_split, epilogue, st)
                  : launch_128<bf16_t, float, false>(A, B, C, R, tail, M, N, K, lda, ldb, ldc, m_split, epilogue, st);
    return glds ? launch_128<bf16_t, bf16_t, true>(A, B, C, R, tail, M, N, K, lda, ldb, ldc, m_split, epilogue, st)
                : launch_128<bf16_t, bf16_t, false>(A, B, C, R, tail, M, N, K, lda, ldb, ldc, m_split, epilogue, st);
  }
  return glds ? launch_128<float, float, true>(A, B, C, R, tail, M, N, K, lda, ldb, ldc, m_split, epilogue, st)
              : launch_128<float, float, false>(A, B, C, R, tail, M, N, K, lda, ldb, ldc, m_split, epilogue, st);
}

// C = A·B^T (+ epilogue) as fvqa_gemm_nt variant 0 computes it, and the small product `rider` — on the CUs the main
// problem leaves idle when the persistent kernel runs it with >= 16 of them to spare, else as its own launch right
// after. Same arithmetic either way (gemm_skinny.h), so results do not depend on where the rider ran.
extern "C" int fvqa_gemm_nt_rider(const void* A, const void* B, void* C, const void* R, int M, int N, int K, int lda,
                                  int ldb, int ldc, int dtype, int out_dtype, int epilogue,
                                  const fvqa_sk_rider* rider, void* workspace, size_t workspace_bytes, void* stream) {
  if (!rider || !rider->A || !rider->B || !rider->C || rider->M <= 0 || rider->N <= 0 || rider->K <= 0) return FVQA_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  const bool swb = epilogue == FVQA_EPI_SWIGLU_BWD || epilogue == FVQA_EPI_SWIGLU_BWD_ST;
  const bool sk_ok = A && B && C && fvqa_dtype_ok(dtype) && (N & 7) == 0 && (ldc & 7) == 0 &&
                     (((uintptr_t)C | (uintptr_t)R) & 15) == 0 && workspace != nullptr &&
                     ((uintptr_t)workspace & 255) == 0 && workspace_bytes >= fvqa_gemm_sk_workspace() &&
                     ((M >= 192 && N >= 256) || swb);
  int rode = 0;
  int rc;
  if (sk_ok) {
    if (epilogue != FVQA_EPI_NONE && epilogue != FVQA_EPI_RESIDUAL && !swb) return FVQA_EINVAL;
    if ((epilogue == FVQA_EPI_RESIDUAL || swb) && (!R || out_dtype != dtype)) return FVQA_EINVAL;
    if (swb && ldc != 2 * N) return FVQA_EINVAL;
    if (out_dtype != dtype && out_dtype != FVQA_F32) return FVQA_EINVAL;
    const int ke = dtype == FVQA_H16 ? 64 : 32;
    const size_t es = fvqa_dtype_size(dtype);
    if (M <= 0 || N <= 0 || K <= 0 || (K % ke) || lda < K || ldb < K || ldc < N) return FVQA_ESHAPE;
    if (((uintptr_t)A & 15) || ((uintptr_t)B & 15) || ((size_t)lda * es & 15) || ((size_t)ldb * es & 15)) return FVQA_EALIGN;
    rc = fvqa_gemm_sk_impl(A, B, C, R, workspace, workspace_bytes, M, N, K, lda, ldb, ldc, dtype, out_dtype, epilogue, st,
                           rider, &rode);
  } else {
    rc = fvqa_gemm_nt(A, B, C, R, nullptr, M, N, K, lda, ldb, ldc, M, dtype, out_dtype, epilogue, 0, workspace,
                      workspace_bytes, stream);
  }
  if (rc || rode) return rc;
  if (rider->accumulate_f32)
    return fvqa_gemm_nt(rider->A, rider->B, nullptr, nullptr, (float*)rider->C, rider->M, rider->N, rider->K, rider->lda,
                        rider->ldb, rider->ldc, 0, dtype, dtype, FVQA_EPI_NONE, 0, nullptr, 0, stream);
  return fvqa_gemm_nt(rider->A, rider->B, rider->C, nullptr, nullptr, rider->M, rider->N, rider->K, rider->lda, rider->ldb,
                      rider->ldc, rider->M, dtype, dtype, FVQA_EPI_NONE, 0, nullptr, 0, stream);
}

extern "C" int fvqa_rope_qk(void* qkv, const float* cos_t, const float* sin_t, int n_seq, int seq_len, int n_heads,
                            int head_dim, int inverse, int dtype, void* stream);

// The QKV projection with RoPE applied to its q | k columns in the epilogue (bf16): C[M, N] = A · B^T, columns [0, rope->cols)
// rotated with the tables of position (row % seq_len). Shapes the persistent kernel does not take run the plain product and
// the row kernel rope_qk_k after it — same arithmetic (value rounded to bf16, rotated in fp32, rounded again).
extern "C" int fvqa_gemm_nt_rope(const void* A, const void* B, void* C, int M, int N, int K, int lda, int ldb, int ldc,
                                 const fvqa_sk_rope* rope, const fvqa_sk_rider* rider, void* workspace,
                                 size_t workspace_bytes, void* stream) {
  if (!A || !B || !C || !rope || !rope->cos_t || !rope->sin_t) return FVQA_EINVAL;
  if (rope->seq_len <= 0 || rope->head_dim <= 0 || (rope->head_dim % 8) || rope->cols <= 0 || rope->cols > N ||
      (rope->cols % (2 * rope->head_dim)) || M % rope->seq_len)
    return FVQA_ESHAPE;
  if (M <= 0 || N <= 0 || K <= 0 || (K % 64) || lda < K || ldb < K || ldc < N) return FVQA_ESHAPE;
  if (((uintptr_t)A & 15) || ((uintptr_t)B & 15) || ((size_t)lda * 2 & 15) || ((size_t)ldb * 2 & 15)) return FVQA_EALIGN;
  hipStream_t st = (hipStream_t)stream;
  const bool sk_ok = (N & 7) == 0 && (ldc & 7) == 0 && ((uintptr_t)C & 15) == 0 && workspace != nullptr &&
                     ((uintptr_t)workspace & 255) == 0 && workspace_bytes >= fvqa_gemm_sk_workspace() && M >= 192 && N >= 256;
  int rode = 0, rc;
  if (sk_ok) {
    rc = fvqa_gemm_sk_impl(A, B, C, nullptr, workspace, workspace_bytes, M, N, K, lda, ldb, ldc, FVQA_H16, FVQA_H16,
                           FVQA_EPI_ROPE, st, rider, &rode, nullptr, rope);
  } else {
    if (ldc != 3 * (rope->cols / 2)) return FVQA_ESHAPE;     // the row kernel takes fused q | k | v rows only
    rc = fvqa_gemm_nt(A, B, C, nullptr, nullptr, M, N, K, lda, ldb, ldc, M, FVQA_H16, FVQA_H16, FVQA_EPI_NONE, 0, workspace,
                      workspace_bytes, stream);
    if (!rc)
      rc = fvqa_rope_qk(C, rope->cos_t, rope->sin_t, M / rope->seq_len, rope->seq_len, rope->cols / 2 / rope->head_dim,
                        rope->head_dim, 0, FVQA_H16, stream);
  }
  if (rc || rode || !rider) return rc;
  if (rider->accumulate_f32)
    return fvqa_gemm_nt(rider->A, rider->B, nullptr, nullptr, (float*)rider->C, rider->M, rider->N, rider->K, rider->lda,
                        rider->ldb, rider->ldc, 0, FVQA_H16, FVQA_H16, FVQA_EPI_NONE, 0, nullptr, 0, stream);
  return fvqa_gemm_nt(rider->A, rider->B, rider->C, nullptr, nullptr, rider->M, rider->N, rider->K, rider->lda, rider->ldb,
                      rider->ldc, rider->M, FVQA_H16, FVQA_H16, FVQA_EPI_NONE, 0, nullptr, 0, stream);
}

static int swiglu_fwd_impl(const void* A, const void* B13, void* ab, void* z, int M, int hidden, int K, int lda, int ldb,
                           int dtype, void* workspace, size_t workspace_bytes, void* stream, int epilogue,
                           const fvqa_sk_rider* rider = nullptr) {
  if (!A || !B13 || !ab || !z || !fvqa_dtype_ok(dtype)) return FVQA_EINVAL;
  const int N = 2 * hidden;
  const int ke = dtype == FVQA_H16 ? 64 : 32;
  const size_t es = fvqa_dtype_size(dtype);
  if (M <= 0 || hidden <= 0 || (hidden % 16) || K <= 0 || (K % ke) || lda < K || ldb < K) return FVQA_ESHAPE;
  if (((uintptr_t)A & 15) || ((uintptr_t)B13 & 15) || ((uintptr_t)ab & 15) || ((uintptr_t)z & 15) ||
      ((size_t)lda * es & 15) || ((size_t)ldb * es & 15) || !workspace || ((uintptr_t)workspace & 255) ||
      workspace_bytes < fvqa_gemm_sk_workspace())
    return FVQA_EALIGN;
  if (rider && (!rider->A || !rider->B || !rider->C || rider->M <= 0 || rider->N <= 0 || rider->K <= 0)) return FVQA_EINVAL;
  int rode = 0;
  const int rc = fvqa_gemm_sk_impl(A, B13, ab, nullptr, workspace, workspace_bytes, M, N, K, lda, ldb, N, dtype, dtype,
                                   epilogue, (hipStream_t)stream, rider, &rode, z);
  if (rc || rode || !rider) return rc;
  if (rider->accumulate_f32)
    return fvqa_gemm_nt(rider->A, rider->B, nullptr, nullptr, (float*)rider->C, rider->M, rider->N, rider->K, rider->lda,
                        rider->ldb, rider->ldc, 0, dtype, dtype, FVQA_EPI_NONE, 0, nullptr, 0, stream);
  return fvqa_gemm_nt(rider->A, rider->B, rider->C, nullptr, nullptr, rider->M, rider->N, rider->K, rider->lda, rider->ldb,
                      rider->ldc, rider->M, dtype, dtype, FVQA_EPI_NONE, 0, nullptr, 0, stream);
}

extern "C" int fvqa_gemm_nt_swiglu_fwd_st_rider(const void* A, const void* B13, void* st, void* z, int M, int hidden, int K,
                                                int lda, int ldb, int dtype, const fvqa_sk_rider* rider, void* workspace,
                                                size_t workspace_bytes, void* stream) {
  return swiglu_fwd_impl(A, B13, st, z, M, hidden, K, lda, ldb, dtype, workspace, workspace_bytes, stream,
                         FVQA_EPI_SWIGLU_FWD_ST, rider);
}

extern "C" int fvqa_gemm_nt_swiglu_fwd(const void* A, const void* B13, void* ab, void* z, int M, int hidden, int K,
                                       int lda, int ldb, int dtype, void* workspace, size_t workspace_bytes,
                                       void* stream) {
  return swiglu_fwd_impl(A, B13, ab, z, M, hidden, K, lda, ldb, dtype, workspace, workspace_bytes, stream,
                         FVQA_EPI_SWIGLU_FWD);
}

extern "C" int fvqa_gemm_nt_swiglu_fwd_st(const void* A, const void* B13, void* st, void* z, int M, int hidden, int K,
                                          int lda, int ldb, int dtype, void* workspace, size_t workspace_bytes,
                                          void* stream) {
  return swiglu_fwd_impl(A, B13, st, z, M, hidden, K, lda, ldb, dtype, workspace, workspace_bytes, stream,
                         FVQA_EPI_SWIGLU_FWD_ST);
}
