// One generated token of the validation / generation path (reference llama/model.py:428-470 re-runs the whole sequence
// through all layers for every new token) as ONE persistent launch: every layer's
//   RMSNorm -> QKV rows -> one-query-row gated attention over the cached K/V -> WO + residual -> RMSNorm -> W1|W3 rows ->
//   SwiGLU -> W2 + residual
// for the M <= 16 new rows (one per sequence), phases separated by grid barriers instead of launch boundaries.
//
// STATUS: opt-in (FVQA_DECODE_PERSISTENT=1), NOT the default token loop. At M = 8 every product is a single pass over its
// weight matrix (HBM-bound, 405 MB per 7B layer) and a launch of the decode-shape GEMM costs ~9 us + bytes / 5.5 TB/s
// (profiles/r03_eval_path.log); the idea was to trade the 5 x 9 us of a layer for 5 grid barriers. Measured on the MI355X
// (profiles/r04_eval_path.log, tools/decode_stamps.py): 200 us per 7B layer against 138 us for the per-kernel sequence —
// a grid barrier over 256 workgroups on 8 XCDs costs 7-8 us (with or without cache write-back / invalidate: it is the drain
// of the write-through stores plus 256 same-address atomics and the poll), one workgroup of 4 waves per CU streams the
// weights at 3.5-4 TB/s where two or three 8-wave workgroups per CU of the stand-alone kernel reach 5.5, and the phases with
// 2.7 (W1|W3 pairs) or 1.5 units per workgroup wait for their slowest member. The per-kernel loop keeps the hardware's
// overlap of one kernel's drain with the next one's ramp for free. Kept because it is exact (bitwise, tested) and is the
// starting point if the barrier gets cheaper (per-XCD counters + flags) — see DESIGN.md section 7.
//
// Arithmetic: bit for bit that of the stand-alone kernels the per-kernel path launches (fvqa/generate.py) —
//   * products: gemm_skinny.h's strip (eight K ranges per 16-column strip, each accumulated in k order by
//     v_mfma_f32_16x16x32_bf16, partial blocks summed in range order, residual added last), in its 256-thread two-strip form;
//   * RMSNorm: rowops.hip rmsnorm_fwd_k (256 threads per row, same per-thread elements, same reduction tree), computed
//     redundantly by every workgroup that needs the normalised rows into its PRIVATE scratch rows (no extra barrier);
//   * SwiGLU: rowops.hip swiglu_fwd_k on the bf16-rounded W1 | W3 rows, as the epilogue of the pair of strips that holds
//     (a, b) of 16 hidden units (AB16 weight order);
//   * attention: attn_decode_body.h, the body of attn_decode_k.
// so tokens are those of the per-kernel path (tests/test_eval.py compares both with the reference's fixtures).
//
// Data between workgroups (the 8 XCDs' L2s are not coherent with each other): every row buffer a phase hands to the next one
// exists ONCE PER LAYER (scratch: x_l, qkv_l, o_l, h_l, z_l, each 256-byte aligned), is written write-through at agent scope
// (sc1 stores: the bytes are in memory when the store is acknowledged) and is read only after the grid barrier that follows its
// phase — no line of it can sit in any cache before it is complete, so plain loads read it and no cache is ever written back
// or invalidated inside the launch (measured with release / acquire fences instead: 8 us per barrier, 5 per layer).
// Grid barrier: every wave drains its stores (s_waitcnt vmcnt(0)), workgroup barrier, then thread 0: one atomic add on a
// counter, bounded poll until all workgroups have arrived, workgroup barrier. All workgroups are resident (grid <= CUs x
// occupancy, checked on the host), every wait is bounded, and a workgroup that times out raises the error word and leaves —
// so do then all others.
#include "attn_decode_body.h"

namespace {

using namespace fvqa_decode;
typedef unsigned long long u64;

constexpr int NPTR = FVQA_DECODE_PTRS;

struct DecArgs {
  const u64* table; int n_layers;
  const bf16_t* x_in; bf16_t* x_out; char* scratch; bf16_t* xn;
  const int32_t* vstart; const int64_t* pos; const float* cs; const float* sn;
  int M, S, H, D, Hf, A, F; float eps; int cache_rot;
  unsigned* bar; u64* err; u64* stamps;
};

enum { E_NONE = 0, E_RESIDUAL = 1, E_SWIGLU = 2 };

__host__ __device__ inline size_t al256(size_t b) { return (b + 255) & ~(size_t)255; }
// per-layer row buffers in `scratch`: x_l | qkv_l | o_l | h_l | z_l
struct LayerBufs { bf16_t* x; bf16_t* qkv; bf16_t* o; bf16_t* h; bf16_t* z; };
__host__ __device__ inline size_t layer_bytes(int M, int D, int Hf) {
  return 3 * al256((size_t)M * D * 2) + al256((size_t)M * 3 * D * 2) + al256((size_t)M * Hf * 2);
}
__device__ __forceinline__ LayerBufs layer_bufs(char* scratch, int l, int M, int D, int Hf) {
  char* p = scratch + (size_t)l * layer_bytes(M, D, Hf);
  LayerBufs b;
  b.x = (bf16_t*)p; p += al256((size_t)M * D * 2);
  b.qkv = (bf16_t*)p; p += al256((size_t)M * 3 * D * 2);
  b.o = (bf16_t*)p; p += al256((size_t)M * D * 2);
  b.h = (bf16_t*)p; p += al256((size_t)M * D * 2);
  b.z = (bf16_t*)p;
  return b;
}

__device__ __forceinline__ void store_wt(bf16_t* p, float v) {       // one bf16, write-through at agent scope
  const unsigned bits = f32_to_bf16_bits(v);
  asm volatile("global_store_short %0, %1, off sc1" ::"v"(p), "v"(bits) : "memory");
}

__device__ __noinline__ bool grid_barrier(unsigned* ctr, unsigned target, u64* err, int* flag) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // this wave's write-through stores are in memory
  __syncthreads();
  if (threadIdx.x == 0) {
    __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    bool ok = false;
    for (unsigned spins = 0; spins < (1u << 21); ++spins) {
      if (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= target) { ok = true; break; }
      if ((spins & 1023) == 1023 && __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) break;
      __builtin_amdgcn_s_sleep(1);
    }
    if (!ok) __hip_atomic_fetch_or(err, (u64)1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    *flag = ok ? 1 : 0;
  }
  __syncthreads();
  return *flag != 0;
}

__device__ __forceinline__ void cvt8(const uint4 t, float (&v)[8]) {
  v[0] = __uint_as_float(t.x << 16); v[1] = __uint_as_float(t.x & 0xFFFF0000u);
  v[2] = __uint_as_float(t.y << 16); v[3] = __uint_as_float(t.y & 0xFFFF0000u);
  v[4] = __uint_as_float(t.z << 16); v[5] = __uint_as_float(t.z & 0xFFFF0000u);
  v[6] = __uint_as_float(t.w << 16); v[7] = __uint_as_float(t.w & 0xFFFF0000u);
}
__device__ __forceinline__ void st8(bf16_t* p, const float (&v)[8]) {
  uint4 t;
  t.x = (unsigned)f32_to_bf16_bits(v[0]) | ((unsigned)f32_to_bf16_bits(v[1]) << 16);
  t.y = (unsigned)f32_to_bf16_bits(v[2]) | ((unsigned)f32_to_bf16_bits(v[3]) << 16);
  t.z = (unsigned)f32_to_bf16_bits(v[4]) | ((unsigned)f32_to_bf16_bits(v[5]) << 16);
  t.w = (unsigned)f32_to_bf16_bits(v[6]) | ((unsigned)f32_to_bf16_bits(v[7]) << 16);
  *reinterpret_cast<uint4*>(p) = t;
}

// rowops.hip rmsnorm_fwd_k for rows 0..M-1 by the 256 threads of this workgroup: every row with that kernel's per-thread elements
// and reduction tree (wave sum, then the four wave sums in order), but the rows side by side — groups of 4 rows are
// loaded together and reduced with ONE pair of workgroup barriers instead of one round trip per row. y: this workgroup's
// private rows (read back by its own strips only).
__device__ __noinline__ void norm_rows(const bf16_t* x, const bf16_t* w, bf16_t* y, int M, int dim, float eps, float (*red8)[4]) {
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  constexpr int RG = 4;                                     // rows per group
  constexpr int KK = 3;                                     // dim <= 6144: up to three 8-element chunks per thread (host-checked)
  for (int r0 = 0; r0 < M; r0 += RG) {
    uint4 raw[RG][KK];
#pragma unroll
    for (int r = 0; r < RG; ++r)
#pragma unroll
      for (int k = 0; k < KK; ++k) {
        const int c = tid * 8 + k * 2048;
        const int rc = r0 + r < M ? r0 + r : M - 1, cc = c < dim ? c : 0;     // (clamped: every load unconditional)
        raw[r][k] = *reinterpret_cast<const uint4*>(x + (size_t)rc * dim + cc);
      }
    float ss[RG];
#pragma unroll
    for (int r = 0; r < RG; ++r) {
      ss[r] = 0.f;
#pragma unroll
      for (int k = 0; k < KK; ++k) {
        float v[8];
        cvt8(raw[r][k], v);
        if (tid * 8 + k * 2048 < dim) {
#pragma unroll
          for (int i = 0; i < 8; ++i) ss[r] += v[i] * v[i];
        }
      }
      ss[r] = wave_sum(ss[r]);
    }
    __syncthreads();
    if (lane == 0) {
#pragma unroll
      for (int r = 0; r < RG; ++r) red8[r][wv] = ss[r];
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < KK; ++k) {
      const int c = tid * 8 + k * 2048;
      if (c < dim) {
        float g[8];
        cvt8(*reinterpret_cast<const uint4*>(w + c), g);
#pragma unroll
        for (int r = 0; r < RG; ++r) {
          if (r0 + r < M) {
            const float tot = red8[r][0] + red8[r][1] + red8[r][2] + red8[r][3];
            const float rr = rsqrtf(tot / (float)dim + eps);
            float v[8], o[8];
            cvt8(raw[r][k], v);
#pragma unroll
            for (int i = 0; i < 8; ++i) o[i] = round_to<bf16_t>(v[i] * rr) * g[i];
            st8(y + (size_t)(r0 + r) * dim + c, o);
          }
        }
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // the rows are read back by this workgroup's strips
  __syncthreads();
}

// NS (1 or 2) adjacent 16-column strips (columns n0 .. n0 + 16 NS - 1) of C = A · B^T by the 4 waves of this workgroup:
// gemm_skinny.h's arithmetic (eight K ranges per strip, wave w takes ranges w and w + 4; every (strip, range) chain
// accumulated in k order; the eight partial blocks summed in range order) with skinny_strip2_4w's pipeline: batches of
// 4 k-steps, double-buffered — 8 NS weight + 8 activation fragments of 16 bytes per lane in flight per buffer. Every load is
// unconditional from a clamped offset (a k-step past the range is zeroed after it arrives: no branch around a load).
template <int EPI, int NS>
__device__ __forceinline__ void strip_run(const bf16_t* A, int lda, const bf16_t* B, int K, int N, int n0, int M, bf16_t* C,
                                          int ldc, const bf16_t* R, float (*part)[8][16][20]) {
  typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
  constexpr int KB = 4, KE = 32 * KB;                       // k-steps / elements per batch
  const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int li = lane & 15, g = lane >> 4;
  const int kw = K / 8;
  const int om = threadIdx.x >> 4, on = threadIdx.x & 15;
  const int am = li < M ? li : M - 1;
  const bf16_t* ap[2];
  const bf16_t* bp[NS][2];
  f32x4 acc[NS][2];
#pragma unroll
  for (int e = 0; e < 2; ++e) {
    ap[e] = A + (size_t)am * lda + (size_t)(w + 4 * e) * kw + 8 * g;
#pragma unroll
    for (int t = 0; t < NS; ++t) {
      int bn = n0 + 16 * t + li; bn = bn < N ? bn : N - 1;
      bp[t][e] = B + (size_t)bn * K + (size_t)(w + 4 * e) * kw + 8 * g;
      acc[t][e] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
  }
  uint4 a0[2][KB], b0[NS][2][KB], a1[2][KB], b1[NS][2][KB];
  auto ld = [&](uint4(&bb)[NS][2][KB], uint4(&aa)[2][KB], int k0) {
#pragma unroll
    for (int e = 0; e < 2; ++e)
#pragma unroll
      for (int u = 0; u < KB; ++u) {
        const int k = k0 + 32 * u;
        const int kc = k < kw ? k : kw - 32;
        aa[e][u] = *reinterpret_cast<const uint4*>(ap[e] + kc);
#pragma unroll
        for (int t = 0; t < NS; ++t) bb[t][e][u] = *reinterpret_cast<const uint4*>(bp[t][e] + kc);
      }
  };
  auto mm = [&](const uint4(&bb)[NS][2][KB], const uint4(&aa)[2][KB], int k0) {
#pragma unroll
    for (int t = 0; t < NS; ++t)
#pragma unroll
      for (int e = 0; e < 2; ++e)
#pragma unroll
        for (int u = 0; u < KB; ++u) {                      // D[n = 4g+r][m = li]; every (strip, range) chain in k order
          const uint4 z4 = make_uint4(0, 0, 0, 0);
          const uint4 av = k0 + 32 * u < kw ? aa[e][u] : z4;                 // (a k-step past the range adds exact zeros)
          acc[t][e] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, bb[t][e][u]),
                                                              __builtin_bit_cast(bf16x8_t, av), acc[t][e], 0, 0, 0);
        }
  };
  ld(b0, a0, 0);
  for (int k0 = 0; k0 < kw; k0 += 2 * KE) {
    ld(b1, a1, k0 + KE);
    mm(b0, a0, k0);
    ld(b0, a0, k0 + 2 * KE);
    mm(b1, a1, k0 + KE);
  }
#pragma unroll
  for (int t = 0; t < NS; ++t)
#pragma unroll
    for (int e = 0; e < 2; ++e)
#pragma unroll
      for (int r = 0; r < 4; ++r) part[t][w + 4 * e][li][4 * g + r] = acc[t][e][r];      // [strip][K range][m][n]
  __syncthreads();
  float v[NS];
#pragma unroll
  for (int t = 0; t < NS; ++t) {
    v[t] = 0.f;
#pragma unroll
    for (int ww = 0; ww < 8; ++ww) v[t] += part[t][ww][om][on];
  }
  if constexpr (EPI == E_SWIGLU) {
    static_assert(NS == 2, "the (a, b) rows of 16 hidden units are a pair of strips");
    // the pair holds (a, b) of hidden units n0/2 .. n0/2+15: rowops.hip swiglu_fwd_k on the rounded rows
    if (om < M && n0 + 16 + on < N) {
      const float a = round_to<bf16_t>(v[0]), b = round_to<bf16_t>(v[1]);
      store_wt(C + (size_t)om * ldc + (n0 >> 1) + on, round_to<bf16_t>(a / (1.f + __expf(-a))) * b);
    }
  } else {
#pragma unroll
    for (int t = 0; t < NS; ++t) {
      const int n = n0 + 16 * t + on;
      if (om < M && n < N) {
        float o = v[t];
        if constexpr (EPI == E_RESIDUAL) o += to_f32<bf16_t>(R[(size_t)om * ldc + n]);
        store_wt(C + (size_t)om * ldc + n, o);
      }
    }
  }
  __syncthreads();                                          // `part` is rewritten by the next strips
}

// Strips of a product phase: the N / 16 strips are dealt out in contiguous, near-equal ranges (workgroup wg of G takes
// [wg T / G, (wg + 1) T / G)); a range is walked in pairs of strips, a single one last. The SwiGLU phase deals out (a, b) pairs.
template <bool PAIRS>
__device__ __forceinline__ void my_range(int N, int wg, int G, int& lo, int& hi) {
  const int T = PAIRS ? (N + 31) / 32 : (N + 15) / 16;
  lo = (int)((long long)wg * T / G); hi = (int)((long long)(wg + 1) * T / G);
  if (PAIRS) { lo *= 2; hi *= 2; }
}
template <int EPI>
__device__ __noinline__ void product_phase(const bf16_t* A, int lda, const bf16_t* B, int K, int N, int M, bf16_t* C, int ldc,
                                           const bf16_t* R, int wg, int G, float (*part)[8][16][20]) {
  constexpr bool PAIRS = EPI == E_SWIGLU;
  int lo, hi;
  my_range<PAIRS>(N, wg, G, lo, hi);
  int i = lo;
  for (; i + 2 <= hi; i += 2) strip_run<EPI, 2>(A, lda, B, K, N, 16 * i, M, C, ldc, R, part);
  if constexpr (!PAIRS) {
    if (i < hi) strip_run<EPI, 1>(A, lda, B, K, N, 16 * i, M, C, ldc, R, part);
  }
}
template <bool PAIRS>
__device__ __forceinline__ bool has_work(int N, int wg, int G) {
  int lo, hi;
  my_range<PAIRS>(N, wg, G, lo, hi);
  return hi > lo;
}

__device__ __noinline__ void attention_phase(const DecArgs& a, const bf16_t* qkv_row, bf16_t* cache, bf16_t* o_row,
                                             const float* g1, const float* g2, int wg, int G, float* sc, float* sa,
                                             float* red, float (*apart)[DH]) {
  for (int it = wg; it < a.M * a.H; it += G) {
    attn_decode_body<bf16_t, true>(qkv_row, cache, o_row, g1, g2, a.vstart, a.pos, a.cs, a.sn, a.M, a.S, a.H, a.A, a.F,
                                   a.cache_rot, it % a.H, it / a.H, sc, sa, red, apart);
    __syncthreads();
  }
}

__global__ __launch_bounds__(256, 2) void decode_token_k(const DecArgs a) {
  __shared__ float sc[SMAX];                                // attention: text scores, then probabilities
  __shared__ float sa[16];
  __shared__ float red[8];
  __shared__ float red8[8][4];
  __shared__ float apart[4][DH];
  __shared__ float part[2][8][16][20];                      // products: partial blocks [strip][K range][m][n]
  __shared__ int bflag;
  // workgroups b and b + G/2 tend to share a CU (dispatch order): give them ADJACENT work ids, so that a CU's two
  // workgroups take neighbouring ranges (1 + 2 strips where a phase has 1.5 per workgroup)
  const int G = gridDim.x;
  const int wg = (G & 1) ? (int)blockIdx.x : ((int)blockIdx.x % (G / 2)) * 2 + (int)blockIdx.x / (G / 2);
  const int M = a.M, D = a.D, Hf = a.Hf;
  bf16_t* xn = a.xn + (size_t)blockIdx.x * M * D;           // this workgroup's private normalised rows
  unsigned target = 0;
  const bf16_t* x = a.x_in;
  for (int l = 0; l < a.n_layers; ++l) {
    const u64* t = a.table + (size_t)l * NPTR;
    const bf16_t* an = (const bf16_t*)t[0];
    const bf16_t* wqkv = (const bf16_t*)t[1];
    const bf16_t* wo = (const bf16_t*)t[2];
    const bf16_t* fn = (const bf16_t*)t[3];
    const bf16_t* w13 = (const bf16_t*)t[4];
    const bf16_t* w2 = (const bf16_t*)t[5];
    bf16_t* cache = (bf16_t*)t[6];
    const float* g1 = (const float*)t[7];
    const float* g2 = (const float*)t[8];
    const LayerBufs b = layer_bufs(a.scratch, l, M, D, Hf);
    bf16_t* xo = l + 1 == a.n_layers ? a.x_out : b.x;
    // (diagnostic: 100 MHz stamps of work id 0's phases in the layer before the last, words 8.. of the workspace)
#define DEC_STAMP(i) do { if (wg == 0 && threadIdx.x == 0 && l == a.n_layers - 2) a.stamps[i] = __builtin_amdgcn_s_memrealtime(); } while (0)
    DEC_STAMP(0);
    // ---- P1: RMSNorm, QKV rows
    if (has_work<false>(3 * D, wg, G)) {
      norm_rows(x, an, xn, M, D, a.eps, red8);
      DEC_STAMP(1);
      product_phase<E_NONE>(xn, D, wqkv, D, 3 * D, M, b.qkv, 3 * D, nullptr, wg, G, part);
    }
    DEC_STAMP(2);
    if (!grid_barrier(a.bar, target += G, a.err, &bflag)) return;
    DEC_STAMP(3);
    // ---- P2: the new rows against the cached keys / values; their k, v join the cache
    attention_phase(a, b.qkv, cache, b.o, g1, g2, wg, G, sc, sa, red, apart);
    DEC_STAMP(4);
    if (!grid_barrier(a.bar, target += G, a.err, &bflag)) return;
    DEC_STAMP(5);
    // ---- P3: WO + residual
    product_phase<E_RESIDUAL>(b.o, D, wo, D, D, M, b.h, D, x, wg, G, part);
    DEC_STAMP(6);
    if (!grid_barrier(a.bar, target += G, a.err, &bflag)) return;
    DEC_STAMP(7);
    // ---- P4: RMSNorm, W1 | W3 rows, SwiGLU
    if (has_work<true>(2 * Hf, wg, G)) {
      norm_rows(b.h, fn, xn, M, D, a.eps, red8);
      DEC_STAMP(8);
      product_phase<E_SWIGLU>(xn, D, w13, D, 2 * Hf, M, b.z, Hf, nullptr, wg, G, part);
    }
    DEC_STAMP(9);
    if (!grid_barrier(a.bar, target += G, a.err, &bflag)) return;
    DEC_STAMP(10);
    // ---- P5: W2 + residual -> the layer's output rows
    product_phase<E_RESIDUAL>(b.z, Hf, w2, Hf, D, M, xo, D, b.h, wg, G, part);
    DEC_STAMP(11);
    if (l + 1 < a.n_layers) {
      if (!grid_barrier(a.bar, target += G, a.err, &bflag)) return;
    }
    DEC_STAMP(12);
    x = b.x;
  }
}

int grid_for_device() {
  int dev = 0, cus = 0, per = 0;
  if (hipGetDevice(&dev) != hipSuccess) return 0;
  if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) return 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per, (const void*)decode_token_k, 256, 0) != hipSuccess || per <= 0) return 0;
  int want = 2;
  if (const char* e = getenv("FVQA_DECODE_WGS_PER_CU")) want = atoi(e);
  if (want < 1) want = 1;
  if (want > per) want = per;
  return cus * want;                                        // every workgroup resident at once: the grid barrier needs that
}

}  // namespace

extern "C" size_t fvqa_decode_workspace(void) { return 256; }

extern "C" int fvqa_decode_token_ok(int n_seq, int seq_len, int n_heads, int head_dim, int hidden, int adapter_len, int dtype) {
  const int D = n_heads * head_dim;
  if (dtype != FVQA_BF16) return 0;
  return n_seq >= 1 && n_seq <= 16 && head_dim == DH && seq_len >= 1 && seq_len <= SMAX && D % 256 == 0 && D <= 6144 &&
         hidden % 256 == 0 && adapter_len >= 0 && adapter_len <= 16 && n_heads <= 65535;
}

extern "C" size_t fvqa_decode_scratch_bytes(int n_layers, int n_seq, int n_heads, int head_dim, int hidden) {
  const int D = n_heads * head_dim;
  int g = grid_for_device();
  if (g <= 0) g = 512;
  return al256((size_t)n_layers * layer_bytes(n_seq, D, hidden)) + al256((size_t)g * n_seq * D * 2);
}

extern "C" int fvqa_decode_token(const uint64_t* table, int n_layers, const void* x, void* x_out, void* scratch, size_t scratch_bytes,
                                 const int32_t* vstart, const int64_t* pos, const float* cos_t, const float* sin_t, int n_seq,
                                 int seq_len, int n_heads, int head_dim, int hidden, int adapter_len, int max_feats, float eps,
                                 int cache_rotated, void* ws, int dtype, void* stream) {
  if (!table || !x || !x_out || !scratch || !vstart || !pos || !cos_t || !sin_t || !ws) return FVQA_EINVAL;
  if (n_layers <= 0 || max_feats < 0) return FVQA_ESHAPE;
  if (!fvqa_decode_token_ok(n_seq, seq_len, n_heads, head_dim, hidden, adapter_len, dtype)) return FVQA_ESHAPE;
  if (((uintptr_t)x | (uintptr_t)x_out | (uintptr_t)ws) & 15) return FVQA_EALIGN;
  if ((uintptr_t)scratch & 255) return FVQA_EALIGN;
  if (scratch_bytes < fvqa_decode_scratch_bytes(n_layers, n_seq, n_heads, head_dim, hidden)) return FVQA_ESHAPE;
  const int G = grid_for_device();
  if (G <= 0) return FVQA_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  if (hipMemsetAsync(ws, 0, 16, st) != hipSuccess) return FVQA_EINVAL;   // the barrier counter (the error word is sticky)
  const int D = n_heads * head_dim;
  DecArgs a;
  a.table = (const u64*)table; a.n_layers = n_layers;
  a.x_in = (const bf16_t*)x; a.x_out = (bf16_t*)x_out; a.scratch = (char*)scratch;
  a.xn = (bf16_t*)((char*)scratch + al256((size_t)n_layers * layer_bytes(n_seq, D, hidden)));
  a.vstart = vstart; a.pos = pos; a.cs = cos_t; a.sn = sin_t;
  a.M = n_seq; a.S = seq_len; a.H = n_heads; a.D = D; a.Hf = hidden; a.A = adapter_len; a.F = max_feats;
  a.eps = eps; a.cache_rot = cache_rotated;
  a.bar = (unsigned*)ws; a.err = (u64*)((char*)ws + 16); a.stamps = (u64*)((char*)ws + 64);
  hipLaunchKernelGGL(decode_token_k, dim3(G), dim3(256), 0, st, a);
  FVQA_CHECK_LAUNCH();
  return FVQA_OK;
}
