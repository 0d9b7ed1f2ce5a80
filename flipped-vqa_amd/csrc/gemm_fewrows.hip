// Few-rows projection: C[M, N] = A[M, K] · B[N, K]^T (+ epilogue) for 17..64 rows against a 7B-class weight matrix — the shapes of
// the TAIL ROWS (include/fvqa.h fvqa_row_segs: the last layer's post-attention half, the LM head and their dX on the few dozen rows a
// head reads). Such a launch is a weight stream: 33-262 MB of B read once, 33 rows of A that stay in L2. The tile kernels of
// gemm_sk.hip / gemm4w.hip keep 64 KiB per workgroup in flight and cut such a problem into 128-250 workgroups (≈ 2.5-3.5 TB/s);
// here the chip is covered with small workgroups that each keep a few KiB in flight:
//
//   fewrows_partial_k  grid (N / 64, KS) x 256 threads. A workgroup owns 64 columns and one of KS ranges of K; its four waves cut
//                      that range four ways again. A wave walks its k-steps in batches of four: per k-step (32 of K) four weight
//                      fragments (4 strips of 16 columns) and RB activation fragments (RB = row blocks of 16) come straight from
//                      global memory into registers — 4·RB MFMAs per (4 + RB) loads — the weight fragment as the MFMA's row operand,
//                      so a lane ends with 4 consecutive columns of one row (as gemm_skinny.h). The four waves' partial blocks meet
//                      in LDS, are summed in wave order and leave as fp32 to partial plane ky of the workspace.
//   fewrows_finish_k   sums the KS planes in plane order and applies the epilogue on whole rows: none (bf16 / fp32 out), + residual,
//                      SwiGLU forward (s, t and z from the a | b columns in AB16 order), SwiGLU' (d(a | b) = dz * (t, s)) — the
//                      arithmetic of the tile kernels' epilogues (gemm4w_kernel.h store_tile4), value for value.
//
// No atomics, fixed summation orders: bitwise repeatable. Replaces F.linear of llama/model.py:127-128, 142, 348 and their dX at the
// row counts of the tail rows.
#include "common.h"
#include "probe.h"
#include <cstdlib>

namespace {

constexpr int FR_U = 4;                       // k-steps per batch

__device__ __forceinline__ size_t fr_ab16(int c) { return (size_t)(c >> 4) * 32 + (c & 15); }

template <int RB>
__global__ __launch_bounds__(256) void fewrows_partial_k(const bf16_t* __restrict__ A, const bf16_t* __restrict__ B,
                                                         float* __restrict__ part, int M, int N, int K, int lda, int ldb,
                                                         int steps) {
  extern __shared__ __attribute__((aligned(16))) float fr_red[];          // [4 waves][RB * 4 blocks][16 n][17]
  const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int li = lane & 15, g = lane >> 4;
  const int n0 = blockIdx.x * 64;
  const int k0 = (blockIdx.y * 4 + w) * steps * 32;
  const int k1 = min(K, k0 + steps * 32);
  const bf16_t* bp[4];
  const bf16_t* ap[RB];
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    int bn = n0 + 16 * s + li; bn = bn < N ? bn : N - 1;
    bp[s] = B + (size_t)bn * ldb + 8 * g;
  }
  f32x4 acc[RB][4];
#pragma unroll
  for (int b = 0; b < RB; ++b) {
    int am = 16 * b + li; am = am < M ? am : M - 1;
    ap[b] = A + (size_t)am * lda + 8 * g;
#pragma unroll
    for (int s = 0; s < 4; ++s) acc[b][s] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  // Batches are walked from a start that differs from workgroup to workgroup (and wave to wave): with K = 4096 the weight rows are
  // 8 KiB apart, so waves that all sit at the same k ask the same few memory channels for everything (measured: the K = 4096 shapes
  // streamed 3.4 TB/s, the K = 11008 / 22016 / 32000 ones 4.5). The sum of a column is still taken in one fixed order per launch.
  const int nb = k1 > k0 ? (k1 - k0 + 32 * FR_U - 1) / (32 * FR_U) : 0;
  const int rot = nb > 0 ? (int)((blockIdx.x * 5u + blockIdx.y * 3u + (unsigned)w) % (unsigned)nb) : 0;
  for (int j = 0; j < nb; ++j) {
    const int bi = j + rot < nb ? j + rot : j + rot - nb;
    const int k = k0 + bi * 32 * FR_U;
    uint4 bf[FR_U][4], af[RB][FR_U];
#pragma unroll
    for (int u = 0; u < FR_U; ++u) {
      const int kk = k + 32 * u;
      const bool in = kk < k1;
#pragma unroll
      for (int s = 0; s < 4; ++s) bf[u][s] = in ? *reinterpret_cast<const uint4*>(bp[s] + kk) : make_uint4(0, 0, 0, 0);
#pragma unroll
      for (int b = 0; b < RB; ++b) af[b][u] = in ? *reinterpret_cast<const uint4*>(ap[b] + kk) : make_uint4(0, 0, 0, 0);
    }
#pragma unroll
    for (int u = 0; u < FR_U; ++u)
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int b = 0; b < RB; ++b)                          // D[n = 16s + 4g + r][m = 16b + li]; every chain in k order
          acc[b][s] = FVQA_MFMA_H16_16x16x32(__builtin_bit_cast(h16x8_t, bf[u][s]), __builtin_bit_cast(h16x8_t, af[b][u]),
                                             acc[b][s], 0, 0, 0);
  }
#pragma unroll
  for (int b = 0; b < RB; ++b)
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int r = 0; r < 4; ++r) fr_red[((w * (RB * 4) + b * 4 + s) * 16 + 4 * g + r) * 17 + li] = acc[b][s][r];
  __syncthreads();
  float* plane = part + (size_t)blockIdx.y * M * N;
  for (int o = threadIdx.x; o < RB * 4 * 256; o += 256) {
    const int blk = o >> 8, nn = o & 15, mm = (o >> 4) & 15;
    const int m = 16 * (blk >> 2) + mm, n = n0 + 16 * (blk & 3) + nn;
    if (m < M && n < N) {
      float v = 0.f;
#pragma unroll
      for (int ww = 0; ww < 4; ++ww) v += fr_red[((ww * (RB * 4) + blk) * 16 + nn) * 17 + mm];
      plane[(size_t)m * N + n] = v;
    }
  }
}

__device__ __forceinline__ f32x4 fr_sum(const float* __restrict__ part, size_t plane, int KS, size_t off) {
  f32x4 v = *reinterpret_cast<const f32x4*>(part + off);
  for (int k = 1; k < KS; ++k) {
    const f32x4 x = *reinterpret_cast<const f32x4*>(part + (size_t)k * plane + off);
    v[0] += x[0]; v[1] += x[1]; v[2] += x[2]; v[3] += x[3];
  }
  return v;
}

template <typename TO>
__device__ __forceinline__ void fr_store4(TO* p, const float (&v)[4]) { Vec4<TO>::store(p, v); }

// EPI: FVQA_EPI_NONE / _RESIDUAL / _SWIGLU_FWD_ST / _SWIGLU_BWD_ST. One thread per 4 output columns of a row (SwiGLU forward: per 4
// hidden units = 4 a-columns and their 4 b-columns).
template <typename TO, int EPI>
__global__ __launch_bounds__(256) void fewrows_finish_k(const float* __restrict__ part, int KS, TO* __restrict__ C,
                                                        const bf16_t* __restrict__ R, bf16_t* __restrict__ C2, int M, int N,
                                                        int ldc) {
  const size_t plane = (size_t)M * N;
  const int per = EPI == FVQA_EPI_SWIGLU_FWD_ST ? N / 8 : N / 4;
  const int t = blockIdx.x * 256 + threadIdx.x;
  const int m = t / per, c = t - m * per;
  if (m >= M) return;
  if constexpr (EPI == FVQA_EPI_SWIGLU_FWD_ST) {
    // columns in AB16 order: blocks (2P, 2P+1) = (a, b) of 16 hidden units. z = silu(a) * b from the values ROUNDED to the storage
    // type (llama/model.py:142); the a / b slots of C get s = silu(a) and t = dz/da = b sigma(a) (1 + a (1 - sigma(a)))
    const int P = c >> 2, q = c & 3;
    const int na = 32 * P + 4 * q;
    const f32x4 a4 = fr_sum(part, plane, KS, (size_t)m * N + na), b4 = fr_sum(part, plane, KS, (size_t)m * N + na + 16);
    float s_[4], t_[4], z_[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float av = round_to<bf16_t>(a4[e]), bv = round_to<bf16_t>(b4[e]);
      const float sg = __builtin_amdgcn_rcpf(1.f + __expf(-av));
      const float sv = round_to<bf16_t>(av * sg);
      z_[e] = sv * bv;
      s_[e] = sv;
      t_[e] = bv * sg * (1.f + av * (1.f - sg));
    }
    fr_store4<bf16_t>((bf16_t*)C + (size_t)m * ldc + na, s_);
    fr_store4<bf16_t>((bf16_t*)C + (size_t)m * ldc + na + 16, t_);
    fr_store4<bf16_t>(C2 + (size_t)m * (N >> 1) + 16 * P + 4 * q, z_);
  } else {
    const int n = 4 * c;
    const f32x4 v4 = fr_sum(part, plane, KS, (size_t)m * N + n);
    float v[4] = {v4[0], v4[1], v4[2], v4[3]};
    if constexpr (EPI == FVQA_EPI_SWIGLU_BWD_ST) {
      // v = dz[m][n..]; R = (s, t) saved by the forward in the a / b slots of the AB16 rows (2N columns), C = d(a | b): da = dz * t,
      // db = dz * s   (llama/model.py:142 backward)
      const size_t o = (size_t)m * ldc + fr_ab16(n);
      float s_[4], t_[4], da[4], db[4];
      Vec4<bf16_t>::load(R + o, s_);
      Vec4<bf16_t>::load(R + o + 16, t_);
#pragma unroll
      for (int e = 0; e < 4; ++e) { da[e] = v[e] * t_[e]; db[e] = v[e] * s_[e]; }
      fr_store4<bf16_t>((bf16_t*)C + o, da);
      fr_store4<bf16_t>((bf16_t*)C + o + 16, db);
    } else {
      if constexpr (EPI == FVQA_EPI_RESIDUAL) {
        float r_[4];
        Vec4<bf16_t>::load(R + (size_t)m * ldc + n, r_);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] += r_[e];
      }
      fr_store4<TO>(C + (size_t)m * ldc + n, v);
    }
  }
}

template <int RB>
int launch_partial(const void* A, const void* B, float* part, int M, int N, int K, int lda, int ldb, int KS, int steps,
                   hipStream_t st) {
  const size_t lds = (size_t)4 * RB * 4 * 16 * 17 * sizeof(float);
  static std::atomic<unsigned long long> attr_done{0};          // one bit per device (fvqa_attr_needed)
  if (fvqa_attr_needed(attr_done))
    (void)hipFuncSetAttribute((const void*)fewrows_partial_k<RB>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL(fewrows_partial_k<RB>, dim3((N + 63) / 64, KS), dim3(256), lds, st, (const bf16_t*)A, (const bf16_t*)B, part,
                     M, N, K, lda, ldb, steps);
  return FVQA_OK;
}

}  // namespace

// K ranges across workgroups (planes of partial sums) and k-steps per wave: >= 3 workgroups per CU where K allows (a wave keeps >= 8
// k-steps), no empty plane
static int fr_planes(int N, int K, int* steps_out) {
  const int groups = (N + 63) / 64, T = K / 32;
  int KS = (768 + groups - 1) / groups;
  const int ks_max = T / 32 > 0 ? T / 32 : 1;
  KS = KS < 1 ? 1 : (KS > 16 ? 16 : KS);
  KS = KS > ks_max ? ks_max : KS;
  const int steps = (T + KS * 4 - 1) / (KS * 4);
  if (steps_out) *steps_out = steps;
  return (T + steps * 4 - 1) / (steps * 4);
}

// 1 when the few-rows kernels take the problem (the dispatcher of gemm_sk.hip asks before planning tiles)
int fvqa_fewrows_takes(int M, int N, int K, int dtype, int out_dtype, int epilogue, size_t scratch_bytes) {
  static const bool off = [] { const char* e = getenv("FVQA_FEWROWS"); return e && e[0] == '0'; }();
  if (off || dtype != FVQA_H16 || M <= 16 || M > 64 || N < 256 || (N % 8) || (K % 32)) return 0;
  if ((size_t)N * (size_t)K < ((size_t)1 << 24)) return 0;
  if (epilogue == FVQA_EPI_NONE) { if (out_dtype != FVQA_H16 && out_dtype != FVQA_F32) return 0; }
  else if (epilogue == FVQA_EPI_RESIDUAL || epilogue == FVQA_EPI_SWIGLU_BWD_ST) { if (out_dtype != FVQA_H16) return 0; }
  else if (epilogue == FVQA_EPI_SWIGLU_FWD_ST) { if (out_dtype != FVQA_H16 || (N % 32)) return 0; }
  else return 0;
  return scratch_bytes >= (size_t)fr_planes(N, K, nullptr) * M * N * sizeof(float) ? 1 : 0;
}

// C = A·B^T (+ epilogue) for 17..64 rows. scratch: >= KS x M x N floats (the slab area of the GEMM workspace); C2: z of the SwiGLU
// forward. The caller has validated pointers, alignment and leading dimensions.
int fvqa_fewrows_impl(const void* A, const void* B, void* C, const void* R, void* C2, float* scratch, int M, int N, int K, int lda,
                      int ldb, int ldc, int out_dtype, int epilogue, hipStream_t st) {
  int steps = 0;
  const int KS = fr_planes(N, K, &steps);
  const int rb = (M + 15) / 16;
  // (measurement probe: the pair of launches as ONE projection; kind bit 8 = few-rows)
  FvqaProbeScope ts(st, 2.0 * M * N * K, epilogue | (out_dtype == FVQA_F32 ? 32 : 0) | 256);
  int rc = FVQA_EINVAL;
  if (rb == 2) rc = launch_partial<2>(A, B, scratch, M, N, K, lda, ldb, KS, steps, st);
  else if (rb == 3) rc = launch_partial<3>(A, B, scratch, M, N, K, lda, ldb, KS, steps, st);
  else if (rb == 4) rc = launch_partial<4>(A, B, scratch, M, N, K, lda, ldb, KS, steps, st);
  if (rc) return rc;
  FVQA_CHECK_LAUNCH();
  const int per = epilogue == FVQA_EPI_SWIGLU_FWD_ST ? N / 8 : N / 4;
  const dim3 grid(((size_t)M * per + 255) / 256), block(256);
#define FIN(TO, EPI)                                                                                                        \
  hipLaunchKernelGGL((fewrows_finish_k<TO, EPI>), grid, block, 0, st, (const float*)scratch, KS, (TO*)C, (const bf16_t*)R, \
                     (bf16_t*)C2, M, N, ldc)
  if (epilogue == FVQA_EPI_NONE) { if (out_dtype == FVQA_F32) FIN(float, FVQA_EPI_NONE); else FIN(bf16_t, FVQA_EPI_NONE); }
  else if (epilogue == FVQA_EPI_RESIDUAL) FIN(bf16_t, FVQA_EPI_RESIDUAL);
  else if (epilogue == FVQA_EPI_SWIGLU_FWD_ST) FIN(bf16_t, FVQA_EPI_SWIGLU_FWD_ST);
  else FIN(bf16_t, FVQA_EPI_SWIGLU_BWD_ST);
#undef FIN
  FVQA_CHECK_LAUNCH();
  return FVQA_OK;
}
