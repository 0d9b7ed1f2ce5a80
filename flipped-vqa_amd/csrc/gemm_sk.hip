// Persistent 256-row projection GEMM for gfx950: a grid of at most one 512-thread workgroup per CU walks whole
// 256 x 256 output tiles with the LDS-DMA ring loop of gemm256_loop.h; where a round of whole tiles would leave more
// than half the CUs idle (N = 4096 outputs: 64 tiles for 256 CUs; the last round of W1|W3) the tiles of that round are
// split along K over 2, 4 or 8 workgroups, which reduce them INSIDE the launch and store the finished tile once, with
// the epilogue (residual add, SwiGLU', fp32 logits) applied there. Replaces, for every F.linear of the step (reference
// llama/model.py:89,127-128,142,348 and their dX), "split-K planes in HBM + a consumer or a fix-up pass that sums
// them". Partition rules: gemm_sk_plan.h.
//
// Hand-off of a partial tile (MI355X_MICROARCH.md, inter-workgroup visibility, first row of the sc1 hand-off table;
// cdna_hip_programming.md G16 R1):
//   producer: every wave stores the accumulator blocks it does not reduce itself to its workgroup's slab with 16-byte
//             WRITE-THROUGH (sc1) stores in register-image order (1 KiB contiguous per wave instruction = whole
//             128-byte lines), s_waitcnt vmcnt(0) in EVERY wave, workgroup barrier, then ONE lane stores the launch
//             epoch to the workgroup's flag with an agent-scope atomic (sc1) store;
//   consumer: one lane per partner polls that partner's flag (relaxed agent-scope sc1 load, bounded spin), workgroup barrier, then
//             every wave reads the partners' slabs with sc1 loads ONLY (they bypass this CU's L1; -DFVQA_SK_ACQUIRE
//             adds the agent-scope acquire fence the general recipe has).
// Results do not depend on dispatch order or XCD placement; every sum runs in piece order 0..s-1, so outputs are
// bitwise repeatable. A workgroup waits only for equal partners that have done the same work before and publish
// before they wait; the grid never exceeds the CU count (160 KiB of LDS = one workgroup per CU), so every workgroup
// is resident. A spin that runs out (~1 s) raises the error word of the workspace and lets the grid drain.
#include "gemm256_loop.h"
#include "gemm_sk_plan.h"
#include "gemm_skinny.h"
#include "gemm4w_kernel.h"
#include "probe.h"
#include <atomic>
#include <cstdlib>

namespace {
using namespace fvqa_ring;

typedef unsigned long long u64;
constexpr int SLAB_FLOATS = 8 * 32 * 256;            // 8 waves x 32 blocks x (64 lanes x 4) = 256 KiB
#ifdef FVQA_SK_NOPACK
constexpr bool SK_PACK24 = false;                       // tuning build: fp32 slabs everywhere (the round-2 exchange)
#else
constexpr bool SK_PACK24 = true;
#endif
constexpr size_t SYNC_BYTES = 4096;                   // u64 words: [0] error, [1 + workgroup] epoch flags (<= 511)

// A second, independent product of at most 16 rows (bf16) that rides on the CUs the main problem leaves idle:
// C2 = A2 · B2^T, strip by strip with the decode-shape routine of gemm_skinny.h (HBM-bound on B2's stream). The step
// uses it for the 10 adapter rows: their K/V projections beside the QKV GEMM (192 of 256 CUs busy) and their
// gradient rows beside the W2^T GEMM (172 busy) — as separate launches they cost 21 us each, 1.4 ms per step.
struct SkRider {
  const void* A; const void* B; void* C;
  int M, N, K, lda, ldb, ldc;
  int acc;               // 0: C (storage dtype) = product; 1: C (fp32) += product
  int n_wg;              // workgroups appended to the grid for it (0: no rider)
};

struct SkArgs {
  const void* A; const void* B; void* C; const void* R;
  void* C2;              // FVQA_EPI_SWIGLU_FWD: z (M, N/2), row stride N/2
  float* slabs; u64* sync; u64* stamps;
  int M, N, K, lda, ldb, ldc;
  u64 epoch;
  fvqa_sk_plan plan;
  SkRider rider;
  // FVQA_EPI_ROPE (the QKV projection): columns [0, rope_cols) are q | k heads of 2*rope_hp dims, row m is position m % rope_S
  const float* rope_cos; const float* rope_sin;
  int rope_S, rope_cols, rope_hp, rope_hmask;
};

__device__ __forceinline__ int xcd_chunk(int bid, int nwg) {     // consecutive work ids share an XCD (bijective)
  const int q = nwg >> 3, r = nwg & 7, x = bid & 7;
  const int base = (x < r) ? x * (q + 1) : r * (q + 1) + (x - r) * q;
  return base + (bid >> 3);
}

__device__ __forceinline__ void unpack8(const uint4& q, float (&v)[8]) {
  v[0] = h16_lo(q.x); v[1] = h16_hi(q.x);
  v[2] = h16_lo(q.y); v[3] = h16_hi(q.y);
  v[4] = h16_lo(q.z); v[5] = h16_hi(q.z);
  v[6] = h16_lo(q.w); v[7] = h16_hi(q.w);
}
__device__ __forceinline__ uint4 pack8(const float (&v)[8]) {
  uint4 t;
  t.x = (unsigned)f32_to_bf16_bits(v[0]) | ((unsigned)f32_to_bf16_bits(v[1]) << 16);
  t.y = (unsigned)f32_to_bf16_bits(v[2]) | ((unsigned)f32_to_bf16_bits(v[3]) << 16);
  t.z = (unsigned)f32_to_bf16_bits(v[4]) | ((unsigned)f32_to_bf16_bits(v[5]) << 16);
  t.w = (unsigned)f32_to_bf16_bits(v[6]) | ((unsigned)f32_to_bf16_bits(v[7]) << 16);
  return t;
}

// Tuning builds (-DFVQA_SK_STAMPS): 100 MHz timestamps of each workgroup's phases, 16 u64 words per workgroup behind the
// slabs (never read by the kernel; tools/sk_check.py prints them). The shipping build compiles none of it.
#ifdef FVQA_SK_STAMPS
#define SK_STAMP(slot) do { if (threadIdx.x == 0 && (slot) < 16) a.stamps[(size_t)wid * 16 + (slot)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define SK_STAMP(slot) do { } while (0)
#endif

// Diagnostic build (-DFVQA_SK_CLOCK, tools/sk_clock.py): the shader clock INSIDE the ring loop, as MI355X_MICROARCH.md
// "DVFS give-back" item 6 prescribes: s_memtime (shader cycles) and s_memrealtime (100 MHz) stamped once before and once
// after the loop of every workgroup's first segment; clock = d(memtime) / d(memrealtime) x 100 MHz. Each launch writes to
// its own slice of a 512-launch ring behind the slabs (memory nothing else in the kernel reads), so a whole step of
// back-to-back launches can be read out afterwards. The shipping build compiles none of it.
#ifdef FVQA_SK_CLOCK
constexpr int CLOCK_RING = 512;
#define SK_CLOCK(t_, r_) asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_), "=s"(r_) :: "memory")
#endif

// column of a[c] in an AB16 row (W1 | W3 projections interleaved in 16-column blocks; b[c] sits 16 columns further on)
__device__ __forceinline__ size_t ab16(int c) { return (size_t)(c >> 4) * 32 + (c & 15); }

// Bounded relaxed poll of one epoch flag by the calling lane.
__device__ __forceinline__ bool wait_epoch(u64* flag, u64 epoch, u64* err) {
  for (unsigned spins = 0; spins < (1u << 22); ++spins) {
    if (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == epoch) return true;
    __builtin_amdgcn_s_sleep(8);
  }
  __hip_atomic_fetch_or(err, (u64)1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return false;
}

// Final store of register blocks [0, n_own) (16 rows each) of every wave's 128 x 64 sub-tile; register block i holds
// tile row block i ^ (rowxor / 16). The accumulators take one round trip through the wave's private 16 KiB of (idle)
// ring LDS so that every store instruction writes whole 128-byte lines: chunk c of row r sits at chunk c ^ (r & 15),
// conflict-free for the fragment writes and the row reads.
//   4-byte outputs: lane k of a row owns columns 4k..4k+3 and 32+4k..32+4k+3; 2-byte outputs: columns 8k..8k+7.
template <typename T, typename TO, int EPI>
__device__ __forceinline__ void store_tile(f32x4 (&acc)[8][4], char* smem, const SkArgs& a, int m0, int n0, int w,
                                           int lane, int n_own, int rowxor) {
  constexpr bool W4 = sizeof(TO) == 4;
  const int wr = w >> 2, wc = w & 3;
  const int crow = lane & 15;
  float* stg = reinterpret_cast<float*>(smem) + w * (64 * 64);
  const int q = lane >> 3, k = lane & 7;
  const int rl = W4 ? ((q & 1) * 8 + (q >> 1)) : q;
  const int cA = W4 ? k : 2 * k, cB = W4 ? 8 + k : 2 * k + 1;
  const int N = a.N, M = a.M, ldc = a.ldc;
  const int nA = n0 + wc * 64 + cA * 4, nB = n0 + wc * 64 + cB * 4;
  const T* R = (const T*)a.R;
  TO* C = (TO*)a.C;
#pragma unroll
  for (int p = 0; p < 2; ++p) {
    if (n_own <= 4 * p) continue;                                       // wave-uniform
    auto row_of = [&](int t) { return W4 ? 16 * (t >> 1) + 4 * (t & 1) + rl : 8 * t + rl; };
    auto live = [&](int t) { return 4 * p + (t >> 1) < n_own; };
    auto grow = [&](int t) { return m0 + wr * 128 + ((p * 64 + row_of(t)) ^ rowxor); };     // output row
    uint4 qa[8], qb[8];
    constexpr bool SWB = EPI == FVQA_EPI_SWIGLU_BWD || EPI == FVQA_EPI_SWIGLU_BWD_ST;
    if constexpr (sizeof(T) == 2 && (SWB || EPI == FVQA_EPI_RESIDUAL)) {
#pragma unroll
      for (int t = 0; t < 8; ++t) {                                   // epilogue operands in flight before the staging
        const int m = grow(t);
        qa[t] = qb[t] = uint4{0u, 0u, 0u, 0u};
        if (live(t) && m < M && nA < N) {
          if constexpr (SWB) {
            const T* rp = R + (size_t)m * ldc + ab16(nA);             // ab rows (AB16): a block, b block 16 columns on
            qa[t] = *reinterpret_cast<const uint4*>(rp);
            qb[t] = *reinterpret_cast<const uint4*>(rp + 16);
          } else {
            qa[t] = *reinterpret_cast<const uint4*>(R + (size_t)m * ldc + nA);
          }
        }
      }
    }
#pragma unroll
    for (int ii = 0; ii < 4; ++ii)
#pragma unroll
      for (int j = 0; j < 4; ++j)
        *reinterpret_cast<f32x4*>(stg + (ii * 16 + crow) * 64 + (((j * 4 + (lane >> 4)) ^ crow) << 2)) = acc[p * 4 + ii][j];
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      const int r = row_of(t);
      const int m = grow(t);
      float v[8];
      {
        const f32x4 lo = *reinterpret_cast<const f32x4*>(stg + r * 64 + ((cA ^ (r & 15)) << 2));
        const f32x4 hi = *reinterpret_cast<const f32x4*>(stg + r * 64 + ((cB ^ (r & 15)) << 2));
        v[0] = lo[0]; v[1] = lo[1]; v[2] = lo[2]; v[3] = lo[3];
        v[4] = hi[0]; v[5] = hi[1]; v[6] = hi[2]; v[7] = hi[3];
      }
      if (!live(t) || m >= M) continue;
      float(&vA)[4] = reinterpret_cast<float(&)[4]>(v[0]);
      float(&vB)[4] = reinterpret_cast<float(&)[4]>(v[4]);
      if constexpr (SWB) {
        // v = dz[m][n..]; R = ab, C = dab (rows of 2N in the AB16 layout): d(silu(a)*b)   (llama/model.py:142 backward)
        // _ST: R holds s = silu(a) and t = d z / d a = b sigma(a) (1 + a (1 - sigma(a))) in the a and b slots (left there by
        // the forward's FVQA_EPI_SWIGLU_FWD_ST epilogue), so da = dz * t and db = dz * s
        const size_t o = (size_t)m * ldc;
        constexpr int hb_ = 16;                                       // b sits 16 columns after a
        const int nA_ = (int)ab16(nA), nB_ = (int)ab16(nB);
        float a_[8], b_[8], da[8], db[8];
        if constexpr (sizeof(T) == 2) {
          if (nA >= N) continue;
          unpack8(qa[t], a_);
          unpack8(qb[t], b_);
        } else {
          if (nA < N) { Vec4<T>::load(R + o + nA_, reinterpret_cast<float(&)[4]>(a_[0]));
                        Vec4<T>::load(R + o + hb_ + nA_, reinterpret_cast<float(&)[4]>(b_[0])); }
          if (nB < N) { Vec4<T>::load(R + o + nB_, reinterpret_cast<float(&)[4]>(a_[4]));
                        Vec4<T>::load(R + o + hb_ + nB_, reinterpret_cast<float(&)[4]>(b_[4])); }
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          if constexpr (EPI == FVQA_EPI_SWIGLU_BWD_ST) {
            da[e] = v[e] * b_[e];
            db[e] = v[e] * a_[e];
          } else {
            const float sg = 1.f / (1.f + __expf(-a_[e]));
            da[e] = v[e] * b_[e] * sg * (1.f + a_[e] * (1.f - sg));
            db[e] = v[e] * a_[e] * sg;
          }
        }
        if constexpr (sizeof(TO) == 2) {
          *reinterpret_cast<uint4*>(C + o + nA_) = pack8(da);
          *reinterpret_cast<uint4*>(C + o + hb_ + nA_) = pack8(db);
        } else {
          if (nA < N) { Vec4<TO>::store(C + o + nA_, reinterpret_cast<float(&)[4]>(da[0]));
                        Vec4<TO>::store(C + o + hb_ + nA_, reinterpret_cast<float(&)[4]>(db[0])); }
          if (nB < N) { Vec4<TO>::store(C + o + nB_, reinterpret_cast<float(&)[4]>(da[4]));
                        Vec4<TO>::store(C + o + hb_ + nB_, reinterpret_cast<float(&)[4]>(db[4])); }
        }
      } else {
        TO* cp = C + (size_t)m * ldc;
        if constexpr (EPI == FVQA_EPI_ROPE && sizeof(TO) == 2) {
          // RoPE of the q | k columns where they are produced (model.py:61-67 applied to wq(x), wk(x)): this lane holds 4
          // rotation pairs of one row; the arithmetic of rope_qk_k / rope8 — value rounded to the storage type, rotated in
          // fp32, rounded again — so the attention kernels read finished operands and no key tile is rotated twice
          if (nA < a.rope_cols) {
            const int pos = m % a.rope_S;
            const size_t ti = (size_t)pos * a.rope_hp + ((nA % (2 * a.rope_hp)) >> 1);
            const float4 c4 = *reinterpret_cast<const float4*>(a.rope_cos + ti);
            const float4 s4 = *reinterpret_cast<const float4*>(a.rope_sin + ti);
            const float cc[4] = {c4.x, c4.y, c4.z, c4.w}, ss[4] = {s4.x, s4.y, s4.z, s4.w};
#pragma unroll
            for (int pr = 0; pr < 4; ++pr) {
              const float e = round_to<TO>(v[2 * pr]), d = round_to<TO>(v[2 * pr + 1]);
              v[2 * pr] = e * cc[pr] - d * ss[pr];
              v[2 * pr + 1] = e * ss[pr] + d * cc[pr];
            }
          }
        }
        if constexpr (EPI == FVQA_EPI_RESIDUAL) {
          float r_[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
          if constexpr (sizeof(T) == 2) {
            unpack8(qa[t], r_);
          } else {
            if (nA < N) Vec4<T>::load(R + (size_t)m * ldc + nA, reinterpret_cast<float(&)[4]>(r_[0]));
            if (nB < N) Vec4<T>::load(R + (size_t)m * ldc + nB, reinterpret_cast<float(&)[4]>(r_[4]));
          }
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] += r_[e];
        }
        if constexpr (sizeof(TO) == 2) {
          if (nA < N) *reinterpret_cast<uint4*>(cp + nA) = pack8(v);
        } else {
          if (nA < N) Vec4<TO>::store(cp + nA, vA);
          if (nB < N) Vec4<TO>::store(cp + nB, vB);
        }
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }
}

// Split-K tile shared by NP workgroups (piece c = this one): publish the register blocks [OWN, 8) of the partial tile,
// fetch blocks of the OWN = 8/NP row blocks this piece reduces from the NP-1 partners, add the NP partials in piece
// order. Thanks to the row permutation (rowxor = c*OWN*16) the reduced rows are ALWAYS register blocks [0, OWN): the
// compiler sees the published accumulators die at the stores, so all (NP-1)*OWN*4 partner loads are in flight at once.
// FVQA_EPI_SWIGLU_FWD: the tile holds the W1 | W3 projections in the AB16 layout, so register blocks j = 0, 2 of a wave
// are a-blocks and j = 1, 3 the matching b-blocks: z = silu(a) * b (llama/model.py:142) is formed lane by lane from the
// values ROUNDED to the storage type (what the separate kernel reads back from `ab`), staged through the wave's LDS like
// the tile itself and stored as 64-byte row segments: 128 rows x 32 z columns per wave.
template <typename T, bool ST>
__device__ __forceinline__ void store_swiglu(f32x4 (&acc)[8][4], char* smem, const SkArgs& a, int m0, int n0, int w,
                                             int lane, int n_own, int rowxor) {
  const int wr = w >> 2, wc = w & 3;
  const int crow = lane & 15;
  float* stg = reinterpret_cast<float*>(smem) + w * (64 * 64);
  T* Z = (T*)a.C2;
  const int M = a.M, NZ = a.N >> 1;
  const int rl = lane >> 2, ch = lane & 3;                          // row within 16, 8-column chunk of the 32
  const int nz = (n0 >> 1) + wc * 32 + ch * 8;
#pragma unroll
  for (int p = 0; p < 2; ++p) {
    if (n_own <= 4 * p) continue;
#pragma unroll
    for (int ii = 0; ii < 4; ++ii)
#pragma unroll
      for (int jz = 0; jz < 2; ++jz) {
        f32x4 z;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float av = round_to<T>(acc[p * 4 + ii][2 * jz][e]), bv = round_to<T>(acc[p * 4 + ii][2 * jz + 1][e]);
          if constexpr (ST) {
            // what the backward needs is not (a, b) but s = silu(a) (d z / d b) and t = d z / d a: they take the a and b
            // slots of the tile, which store_tile writes AFTER this pass (the SwiGLU' epilogue is then two multiplies)
            const float sg = __builtin_amdgcn_rcpf(1.f + __expf(-av));
            const float sv = round_to<T>(av * sg);
            z[e] = sv * bv;
            acc[p * 4 + ii][2 * jz][e] = sv;
            acc[p * 4 + ii][2 * jz + 1][e] = bv * sg * (1.f + av * (1.f - sg));
          } else {
            z[e] = round_to<T>(av / (1.f + __expf(-av))) * bv;
          }
        }
        // row (ii*16 + crow) of the pass, z columns jz*16 + 4*(lane>>4) ..+3; chunk c of a row sits at c ^ (row & 7)
        *reinterpret_cast<f32x4*>(stg + (ii * 16 + crow) * 32 + (((jz * 4 + (lane >> 4)) ^ (crow & 7)) << 2)) = z;
      }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int r = t * 16 + rl;
      const int m = m0 + wr * 128 + ((p * 64 + r) ^ rowxor);
      float v[8];
      const f32x4 lo = *reinterpret_cast<const f32x4*>(stg + r * 32 + (((2 * ch) ^ (r & 7)) << 2));
      const f32x4 hi = *reinterpret_cast<const f32x4*>(stg + r * 32 + (((2 * ch + 1) ^ (r & 7)) << 2));
      v[0] = lo[0]; v[1] = lo[1]; v[2] = lo[2]; v[3] = lo[3];
      v[4] = hi[0]; v[5] = hi[1]; v[6] = hi[2]; v[7] = hi[3];
      if (4 * p + (t) >= n_own || m >= M || nz >= NZ) continue;
      if constexpr (sizeof(T) == 2) {
        *reinterpret_cast<uint4*>(Z + (size_t)m * NZ + nz) = pack8(v);
      } else {
        Vec4<T>::store(Z + (size_t)m * NZ + nz, reinterpret_cast<float(&)[4]>(v[0]));
        Vec4<T>::store(Z + (size_t)m * NZ + nz + 4, reinterpret_cast<float(&)[4]>(v[4]));
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }
}

// 24-bit slab format (PACK: launches whose tile is stored as bf16). A partial sum crosses the fabric with sign, the full
// 8-bit exponent and the top 15 mantissa bits (round to nearest on the dropped byte): relative error 2^-17 per partial,
// 1/256 of the half-ulp of the bf16 value the finished tile is rounded to — and a quarter fewer bytes on an exchange
// that is bound by the fabric (stamps: 48 MB out + 48 MB in per N = 4096 launch at ~8.6 TB/s each way). A row block of a
// wave (4 accumulator quads = 16 values per lane) packs into three 16-byte words: each quad's first three values keep
// their top three bytes, the freed low bytes carry the fourth value. Deterministic, so results stay bitwise repeatable;
// the workgroup's own partial enters the sum unrounded. fp32-output launches exchange plain fp32.
// A non-finite partial must stay non-finite (the loss scaler's overflow check sees the reduced tile): the rounding increment
// is skipped when the exponent is all ones — added to a NaN whose mantissa bits 7..22 are all set it would carry into the
// exponent and sign and leave -0 or a tiny finite value. (Inf stays Inf, a NaN stays a NaN or, if only its low byte was set,
// becomes Inf.)
__device__ __forceinline__ unsigned round24(float f) {
  const unsigned u = __float_as_uint(f);
  return (u & 0x7F800000u) == 0x7F800000u ? u : u + 0x80u;
}
__device__ __forceinline__ void pack24(const f32x4& q, unsigned (&d)[3]) {
  const unsigned u0 = round24(q[0]), u1 = round24(q[1]), u2 = round24(q[2]), u3 = round24(q[3]);
  d[0] = __builtin_amdgcn_perm(u0, u3, 0x07060503u);     // bytes: u3.b3 | u0.b1 u0.b2 u0.b3
  d[1] = __builtin_amdgcn_perm(u1, u3, 0x07060502u);     //        u3.b2 | u1.b1..b3
  d[2] = __builtin_amdgcn_perm(u2, u3, 0x07060501u);     //        u3.b1 | u2.b1..b3
}
__device__ __forceinline__ f32x4 unpack24(unsigned d0, unsigned d1, unsigned d2) {
  const unsigned t = __builtin_amdgcn_perm(d0, d1, 0x04000C0Cu);      // d0.b0 -> b3, d1.b0 -> b2, zeros below
  const unsigned u3 = __builtin_amdgcn_perm(t, d2, 0x0706000Cu);      // keep b3, b2; d2.b0 -> b1; b0 = 0
  return f32x4{__uint_as_float(d0 & 0xFFFFFF00u), __uint_as_float(d1 & 0xFFFFFF00u), __uint_as_float(d2 & 0xFFFFFF00u),
               __uint_as_float(u3)};
}

template <int NP, bool PACK>
__device__ __forceinline__ void exchange_reduce(f32x4 (&acc)[8][4], const SkArgs& a, int wid, int c, int g, int jm,
                                                int w, int lane, int tid) {
  constexpr int OWN = 8 / NP;
  constexpr int WPB = PACK ? 3 : 4;                       // 16-byte words per lane per row block in a slab
  const unsigned lane_off = (unsigned)(w * 32 * 1024 + lane * 16);   // this lane's 16 bytes of word 0 of block 0 in a slab
  {
    float* my = a.slabs + (size_t)wid * SLAB_FLOATS;
    const auto rs = __builtin_amdgcn_make_buffer_rsrc(my, 0, SLAB_FLOATS * 4, 0x00020000);
#pragma unroll
    for (int i = OWN; i < 8; ++i) {
      if constexpr (PACK) {
        unsigned d[4][3];
#pragma unroll
        for (int j = 0; j < 4; ++j) pack24(acc[i][j], d[j]);
        const u32x4 w0 = u32x4{d[0][0], d[0][1], d[0][2], d[1][0]}, w1 = u32x4{d[1][1], d[1][2], d[2][0], d[2][1]},
                    w2 = u32x4{d[2][2], d[3][0], d[3][1], d[3][2]};
        __builtin_amdgcn_raw_buffer_store_b128(w0, rs, lane_off + (unsigned)((i * 3 + 0) * 1024), 0, 16);
        __builtin_amdgcn_raw_buffer_store_b128(w1, rs, lane_off + (unsigned)((i * 3 + 1) * 1024), 0, 16);
        __builtin_amdgcn_raw_buffer_store_b128(w2, rs, lane_off + (unsigned)((i * 3 + 2) * 1024), 0, 16);
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j)
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, acc[i][j]), rs,
                                                 lane_off + (unsigned)((i * 4 + j) * 1024), 0, 16);   // aux 16 = sc1
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // EVERY storing wave drains its write-through stores
  }
  __syncthreads();                                        // (also: every wave is done reading the ring)
  SK_STAMP(2);
  const int ts = a.plan.ts;
  if (tid < NP - 1) {                   // lanes 0 .. NP-2 of wave 0: one partner's flag each, polled side by side
    if (tid == 0) __hip_atomic_store(a.sync + 1 + wid, a.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int p = tid < c ? tid : tid + 1;                // (a lost partner raises the error word after its bounded wait)
    (void)wait_epoch(a.sync + 1 + fvqa_sk_piece_team(a.plan, g, c, p) * ts + jm, a.epoch, a.sync);
#ifdef FVQA_SK_ACQUIRE
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");  // not needed while EVERY load of a partner's slab is an sc1 load
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#else
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");   // (no instruction: keeps the loads below the poll)
#endif
  }
  __syncthreads();
  SK_STAMP(3);
  u32x4 x[NP - 1][OWN * WPB];
#pragma unroll
  for (int q = 0; q < NP - 1; ++q) {
    const int p = q < c ? q : q + 1;                      // partner piece
    const float* sl = a.slabs + (size_t)(fvqa_sk_piece_team(a.plan, g, c, p) * ts + jm) * SLAB_FLOATS;
    const auto rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(sl), 0, SLAB_FLOATS * 4, 0x00020000);
    // my tile row block c*OWN + i sits in partner p's register block ((c ^ p) * OWN + i)
    const unsigned boff = lane_off + (unsigned)(((c ^ p) * OWN) * WPB * 1024);
#pragma unroll
    for (int b = 0; b < OWN * WPB; ++b) x[q][b] = __builtin_amdgcn_raw_buffer_load_b128(rs, boff + (unsigned)(b * 1024), 0, 16);
  }
  if constexpr (PACK) {
#pragma unroll
    for (int i = 0; i < OWN; ++i) {
      f32x4 sum[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) sum[j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int q = 0; q < NP - 1; ++q) {                  // pieces in K order: 0, 1, ..., NP-1 (own partial at position c)
        if (q == c) {
#pragma unroll
          for (int j = 0; j < 4; ++j) sum[j] += acc[i][j];
        }
        const u32x4 w0 = x[q][i * 3], w1 = x[q][i * 3 + 1], w2 = x[q][i * 3 + 2];
        sum[0] += unpack24(w0[0], w0[1], w0[2]);
        sum[1] += unpack24(w0[3], w1[0], w1[1]);
        sum[2] += unpack24(w1[2], w1[3], w2[0]);
        sum[3] += unpack24(w2[1], w2[2], w2[3]);
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = (c == NP - 1) ? sum[j] + acc[i][j] : sum[j];
    }
  } else {
#pragma unroll
    for (int b = 0; b < OWN * 4; ++b) {
      f32x4 sum = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int q = 0; q < NP - 1; ++q) {                  // pieces in K order: 0, 1, ..., NP-1 (own partial at position c)
        if (q == c) sum += acc[b >> 2][b & 3];
        sum += __builtin_bit_cast(f32x4, x[q][b]);
      }
      if (c == NP - 1) sum += acc[b >> 2][b & 3];
      acc[b >> 2][b & 3] = sum;
    }
  }
}

template <typename T, typename TO, int EPI>
__global__ __launch_bounds__(512) void gemm_sk_256(const SkArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int KE = Mma<T>::KE;
  const fvqa_sk_plan& P = a.plan;
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n_main = P.n_teams * P.ts;
  if ((int)blockIdx.x >= n_main) {                        // rider workgroups (spread over the XCDs by the dispatcher)
    if constexpr (sizeof(T) == 2) {
      const SkRider& r = a.rider;
      float(*part)[16][20] = reinterpret_cast<float(*)[16][20]>(smem);
      for (int strip = (int)blockIdx.x - n_main; strip * 16 < r.N; strip += r.n_wg) {
        if (r.acc)
          skinny_strip<float, FVQA_EPI_SKINNY_ACC>((const bf16_t*)r.A, (const bf16_t*)r.B, (float*)r.C, nullptr, r.M, r.N,
                                                   r.K, r.lda, r.ldb, r.ldc, strip * 16, part);
        else
          skinny_strip<bf16_t, FVQA_EPI_NONE>((const bf16_t*)r.A, (const bf16_t*)r.B, (bf16_t*)r.C, nullptr, r.M, r.N,
                                              r.K, r.lda, r.ldb, r.ldc, strip * 16, part);
        __syncthreads();                                  // `part` is rewritten by the next strip
      }
    }
    return;
  }
  const int wid = xcd_chunk(blockIdx.x, n_main);
  const int g = wid / P.ts, jm = wid - g * P.ts;
  SK_STAMP(0);
  fvqa_sk_seg s;
  for (int idx = 0; fvqa_sk_segment(P, g, idx, &s); ++idx) {
    const int tni = s.tile / P.mgroups, mg = s.tile - tni * P.mgroups;
    const int mt = mg * P.ts + jm;
    if (mt >= P.tm) continue;                             // m group with fewer tiles than the team has members
    const int m0 = mt * TM, n0 = tni * 256;
    const int own = 8 / s.n;                              // register blocks this piece reduces and stores
    const int rowxor = s.n > 1 ? s.c * own * 16 : 0;

    f32x4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#ifdef FVQA_SK_CLOCK
    u64 ct0, cr0, ct1, cr1;
    __builtin_amdgcn_sched_barrier(0);
    SK_CLOCK(ct0, cr0);
    __builtin_amdgcn_sched_barrier(0);
#endif
    ring_loop<T, 4>(acc, smem, (const T*)a.A, (const T*)a.B, a.M, a.N, a.lda, a.ldb, m0, n0,
                    (size_t)s.k0 * 2 * KE, s.k1 - s.k0, w, lane, rowxor);
#ifdef FVQA_SK_CLOCK
    __builtin_amdgcn_sched_barrier(0);
    SK_CLOCK(ct1, cr1);
    __builtin_amdgcn_sched_barrier(0);
    if (tid == 0 && idx == 0) {
      u64* st_ = a.stamps + (size_t)wid * 16;
      st_[0] = ct0; st_[1] = cr0; st_[2] = ct1; st_[3] = cr1;
      st_[4] = ((u64)(unsigned)a.N << 32) | (unsigned)a.K;
      st_[5] = a.epoch;
      st_[6] = (u64)EPI | ((u64)s.n << 8) | ((u64)(s.k1 - s.k0) << 16) | ((u64)sizeof(TO) << 40) | ((u64)n_main << 48);
      st_[7] = (u64)a.M;
    }
#endif
    SK_STAMP(1);
    // Everything below is addressed from opaque copies of (lane, wave, tile origin): otherwise hipcc hoists the
    // epilogue's per-lane address arithmetic above the ring loop and pays for it with spills INSIDE that loop.
    int lane_e = lane, w_e = w, m0_e = m0, n0_e = n0;
    asm volatile("" : "+v"(lane_e), "+s"(w_e), "+s"(m0_e), "+s"(n0_e));
    const int tid_e = w_e * 64 + lane_e;
    if (s.n == 1) __syncthreads();                        // every wave is done reading the ring
    else {
      constexpr bool PK = SK_PACK24 && sizeof(TO) == 2;  // tiles rounded to bf16 at the store: 24-bit slabs
      if (s.n == 2) exchange_reduce<2, PK>(acc, a, wid, s.c, g, jm, w_e, lane_e, tid_e);
      else if (s.n == 4) exchange_reduce<4, PK>(acc, a, wid, s.c, g, jm, w_e, lane_e, tid_e);
      else exchange_reduce<8, PK>(acc, a, wid, s.c, g, jm, w_e, lane_e, tid_e);
    }
    SK_STAMP(4);
    if constexpr (EPI == FVQA_EPI_SWIGLU_FWD) {
      store_tile<T, TO, FVQA_EPI_NONE>(acc, smem, a, m0_e, n0_e, w_e, lane_e, own, rowxor);      // ab (saved for the backward)
      store_swiglu<T, false>(acc, smem, a, m0_e, n0_e, w_e, lane_e, own, rowxor);                 // z
    } else if constexpr (EPI == FVQA_EPI_SWIGLU_FWD_ST) {
      store_swiglu<T, true>(acc, smem, a, m0_e, n0_e, w_e, lane_e, own, rowxor);                  // z; acc <- (s, t)
      store_tile<T, TO, FVQA_EPI_NONE>(acc, smem, a, m0_e, n0_e, w_e, lane_e, own, rowxor);      // st (saved for the backward)
    } else {
      store_tile<T, TO, EPI>(acc, smem, a, m0_e, n0_e, w_e, lane_e, own, rowxor);
    }
    __syncthreads();                                      // staging reads done before the next segment's DMA
    SK_STAMP(5);
  }
}


// ---- the same partition and hand-off protocol on the 4-wave main loop (gemm4w_loop.h: one wave per SIMD, a wave = 64 rows x 256
// columns, accumulators acc[j][4 i + e] for row block i < 4 and column block j < 16): the loop of a split piece is 7-10 % faster
// than the 8-wave one above (W1|W3^T piece, 86 stages: 115.9 against 129.4 us; profiles/r04_gemm4w.log). Pieces: 2 or 4 per
// tile (a wave has 4 row blocks to share out); bf16 tiles, 24-bit slabs. The WHOLE hand-off — publish, barrier, flag and
// bounded poll, barrier, fetch-and-add in piece order — is generated assembly inside the loop's own statement (gemm4w_asm.h
// Ring4AsmSK): with C++ anywhere between the loop and the last touch of the accumulators hipcc spilled them to scratch
// (226-754 registers). This kernel only prepares the parameters: one register, parameter k in lane k. Slab image of a
// workgroup: wave w at w * 64 KiB, row block i at i * 12 KiB, its 16 quads as 12 words of 1 KiB.
template <typename TO, int EPI>
__global__ __launch_bounds__(256) void gemm4w_sk_k(const SkArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  using X = fvqa_ring4::Ring4AsmSK;
  const fvqa_sk_plan& P = a.plan;
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
  const int n_main = P.n_teams * P.ts;
  if ((int)blockIdx.x >= n_main) {                        // rider workgroups (spread over the XCDs by the dispatcher)
    const SkRider& r = a.rider;
    float(*part)[8][16][20] = reinterpret_cast<float(*)[8][16][20]>(smem);
    for (int pair = (int)blockIdx.x - n_main; pair * 32 < r.N; pair += r.n_wg) {
      if (r.acc)
        skinny_strip2_4w<float, FVQA_EPI_SKINNY_ACC>((const bf16_t*)r.A, (const bf16_t*)r.B, (float*)r.C, r.M, r.N, r.K, r.lda,
                                                     r.ldb, r.ldc, pair * 32, part);
      else
        skinny_strip2_4w<bf16_t, FVQA_EPI_NONE>((const bf16_t*)r.A, (const bf16_t*)r.B, (bf16_t*)r.C, r.M, r.N, r.K, r.lda,
                                                r.ldb, r.ldc, pair * 32, part);
      __syncthreads();
    }
    return;
  }
  const int wid = xcd_chunk(blockIdx.x, n_main);
  const int g = wid / P.ts, jm = wid - g * P.ts;
  auto slab48 = [&](int id) { return (unsigned long long)(uintptr_t)(a.slabs + (size_t)id * SLAB_FLOATS) & 0xFFFFFFFFFFFFull; };
  fvqa_sk_seg s;
  for (int idx = 0; fvqa_sk_segment(P, g, idx, &s); ++idx) {
    const int tni = s.tile / P.mgroups, mg = s.tile - tni * P.mgroups;
    const int mt = mg * P.ts + jm;
    if (mt >= P.tm) continue;                             // m group with fewer tiles than the team has members
    const int m0 = mt * TM, n0 = tni * 256;
    const int own = 4 / s.n;                              // register row blocks this piece reduces and stores
    const int rowxor = s.n > 1 ? s.c * own * 16 : 0;
    // exchange parameters, parameter k in lane k; partners in ascending piece order with this piece left out
    auto partner_wid = [&](int q) { const int p = q < s.c ? q : q + 1; return fvqa_sk_piece_team(P, g, s.c, p) * P.ts + jm; };
    unsigned xp = 0;
    {
      const int k = lane;
      const unsigned long long mine = slab48(wid), fl = (unsigned long long)(uintptr_t)(a.sync + 1 + wid),
                               er = (unsigned long long)(uintptr_t)a.sync;
      if (k == X::XP_NP) xp = (unsigned)s.n;
      else if (k == X::XP_C) xp = (unsigned)s.c;
      else if (k == X::XP_SLAB) xp = (unsigned)mine;
      else if (k == X::XP_SLAB + 1) xp = (unsigned)(mine >> 32);
      else if (k >= X::XP_PSLAB && k < X::XP_PSLAB + 6) {
        const int q = (k - X::XP_PSLAB) >> 1;
        if (q < s.n - 1) {
          const unsigned long long b = slab48(partner_wid(q));
          xp = (k - X::XP_PSLAB) & 1 ? (unsigned)(b >> 32) : (unsigned)b;
        }
      } else if (k >= X::XP_POFF && k < X::XP_POFF + 3) {
        const int q = k - X::XP_POFF, p = q < s.c ? q : q + 1;
        xp = (unsigned)(((s.c ^ p) * own) * 12 * 1024);   // my tile row block c*own + i sits in partner p's block ((c ^ p) * own + i)
      } else if (k == X::XP_FLAG) xp = (unsigned)fl;
      else if (k == X::XP_FLAG + 1) xp = (unsigned)(fl >> 32);
      else if (k == X::XP_EPOCH) xp = (unsigned)a.epoch;
      else if (k == X::XP_EPOCH + 1) xp = (unsigned)(a.epoch >> 32);
      else if (k == X::XP_ERR) xp = (unsigned)er;
      else if (k == X::XP_ERR + 1) xp = (unsigned)(er >> 32);
      else if (k == X::XP_WOFF) xp = (unsigned)(w * 64 * 1024);
      else if (k == X::XP_SLABBYTES) xp = (unsigned)(SLAB_FLOATS * 4);
#ifdef FVQA_SK_CLOCK
      const unsigned long long sa = idx == 0 ? (unsigned long long)(uintptr_t)(a.stamps + (size_t)wid * 16) : 0ull;
      if (k == X::XP_STAMP) xp = (unsigned)sa;
      else if (k == X::XP_STAMP + 1) xp = (unsigned)(sa >> 32);
#endif
    }
    const unsigned long long fa = (unsigned long long)(uintptr_t)(a.sync + 1 + (lane < s.n - 1 ? partner_wid(lane) : wid));
    fvqa_ring4::f32x16 acc[16];
#pragma unroll
    for (int j = 0; j < 16; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;
    fvqa_ring4::ring4_loop_sk(acc, lds0, (const bf16_t*)a.A, (const bf16_t*)a.B, a.M, a.N, a.lda, a.ldb, m0, n0,
                              (size_t)s.k0 * 64, s.k1 - s.k0, w, lane, rowxor, xp, fa);
#ifdef FVQA_SK_CLOCK
    if (tid == 0 && idx == 0) {                           // (words 0-3 of the record: the statement's own stamps around its loop)
      u64* st_ = a.stamps + (size_t)wid * 16;
      st_[4] = ((u64)(unsigned)a.N << 32) | (unsigned)a.K;
      st_[5] = a.epoch;
      st_[6] = (u64)EPI | ((u64)s.n << 8) | ((u64)(s.k1 - s.k0) << 16) | (16ull << 32) | ((u64)sizeof(TO) << 40) | ((u64)n_main << 48);
      st_[7] = (u64)a.M;
    }
#endif
    int lane_e = lane, w_e = w, m0_e = m0, n0_e = n0;      // (see gemm_sk_256: keeps the epilogue's addressing below the loop)
    asm volatile("" : "+v"(lane_e), "+s"(w_e), "+s"(m0_e), "+s"(n0_e));
    fvqa_g4::store_tile4<16, TO, EPI>(acc, smem, a, m0_e, n0_e, w_e, lane_e, own, rowxor);
    __syncthreads();                                      // staging reads done before the next segment's DMA
#ifdef FVQA_SK_CLOCK
    if (tid == 0 && idx == 0) (a.stamps + (size_t)wid * 16)[14] = __builtin_amdgcn_s_memrealtime();   // tile stored
#endif
  }
}

template <typename TO, int EPI>
int launch_sk4(const SkArgs& a, hipStream_t st) {
  auto k = gemm4w_sk_k<TO, EPI>;
  static std::atomic<unsigned long long> attr_done{0};          // one bit per device (fvqa_attr_needed)
  if (fvqa_attr_needed(attr_done)) {
    (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, fvqa_ring4::Geo<16>::RING_BYTES);
  }
  {
    FvqaProbeScope ts(st, 2.0 * a.M * a.N * a.K, EPI | (a.plan.s > 1 ? 16 : 0) | (sizeof(TO) == 4 ? 32 : 0) | 128);
    hipLaunchKernelGGL(k, dim3(a.plan.n_teams * a.plan.ts + a.rider.n_wg), dim3(256), fvqa_ring4::Geo<16>::RING_BYTES, st, a);
  }
  FVQA_CHECK_LAUNCH();
  return FVQA_OK;
}

// Launch epochs of THIS library. libfvqa_hip.so and libfvqa_hip_f16.so may serve one process (and then share the stream's
// workspace, its flags and its error word): flags are compared for EQUALITY with the launch's epoch, so disjoint ranges keep
// a flag value left by the other library from ever matching.
#ifdef FVQA_H16_F16
std::atomic<unsigned long long> g_epoch{1ull << 62};
#else
std::atomic<unsigned long long> g_epoch{0};
#endif

template <typename T, typename TO, int EPI>
int launch_sk(const SkArgs& a, hipStream_t st) {
  auto k = gemm_sk_256<T, TO, EPI>;
  static std::atomic<unsigned long long> attr_done{0};          // one bit per device (fvqa_attr_needed)
  if (fvqa_attr_needed(attr_done)) {
    (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, RING_BYTES);
  }
  {
    FvqaProbeScope ts(st, 2.0 * a.M * a.N * a.K, EPI | (a.plan.s > 1 ? 16 : 0) | (sizeof(TO) == 4 ? 32 : 0) | (sizeof(T) == 4 ? 64 : 0));
    hipLaunchKernelGGL(k, dim3(a.plan.n_teams * a.plan.ts + a.rider.n_wg), dim3(512), RING_BYTES, st, a);
  }
  FVQA_CHECK_LAUNCH();
  return FVQA_OK;
}

int cu_count() {                                          // of the CURRENT device (cached per device id)
  constexpr int MAXDEV = 64;
  static std::atomic<int> n[MAXDEV];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MAXDEV) return 256;
  const int v = n[dev].load();
  if (v > 0) return v;
  int cus = 0;
  if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0)
    return 256;                                           // MI355X; also what the host-only plan queries assume
  n[dev].store(cus);
  return cus;
}

}  // namespace

#ifdef FVQA_SK_CLOCK
constexpr size_t STAMP_BYTES = (size_t)CLOCK_RING * 256 * 16 * sizeof(u64);
#else
constexpr size_t STAMP_BYTES = 256 * 16 * sizeof(u64);
#endif
extern "C" size_t fvqa_gemm_sk_workspace(void) {
  return SYNC_BYTES + (size_t)256 * SLAB_FLOATS * sizeof(float) + STAMP_BYTES;
}

extern "C" int fvqa_gemm_sk_describe(int M, int N, int K, int dtype, int n_cu, int32_t* plan_out, int team,
                                     int32_t* segs_out, int max_segs) {
  if (M <= 0 || N <= 0 || K <= 0 || !fvqa_dtype_ok(dtype) || n_cu <= 0) return FVQA_EINVAL;
  const int wide = dtype == FVQA_H16 ? 64 : 32;
  if (K % wide) return FVQA_ESHAPE;
  const fvqa_sk_plan p = fvqa_sk_make_plan(M, N, K, wide, n_cu);
  if (plan_out) {
    const int32_t v[12] = {p.tm, p.tn, p.nw_tile, p.gran, p.gpt, p.ts, p.mgroups, p.n_teams, p.full, p.rem, p.s, p.pstride};
    for (int i = 0; i < 12; ++i) plan_out[i] = v[i];
  }
  int n = 0;
  if (team >= 0 && team < p.n_teams) {
    fvqa_sk_seg s;
    for (int idx = 0; fvqa_sk_segment(p, team, idx, &s); ++idx) {
      if (segs_out && n < max_segs) {
        const int32_t v[5] = {s.tile, s.k0, s.k1, s.n, s.c};
        for (int i = 0; i < 5; ++i) segs_out[n * 5 + i] = v[i];
      }
      ++n;
    }
  }
  return n;
}

// the 4-wave whole-tile kernel (gemm4w.hip)
extern "C" int fvqa_gemm4w_choose(int M, int N, int K, int dtype, int out_dtype, int epilogue, const fvqa_sk_rider* rider,
                                  int n_cu);
int fvqa_gemm4w_impl(int nbt, const void* A, const void* B, void* C, const void* R, int M, int N, int K, int lda, int ldb,
                     int ldc, int out_dtype, int epilogue, hipStream_t st, const fvqa_sk_rider* rider, int* rode, void* C2,
                     const fvqa_sk_rope* rope, int n_cu, unsigned long long* clock_stamps, unsigned long long epoch);

// C[M,N] = A[M,K] x B[N,K]^T with the epilogue applied once per finished tile. `ws`: fvqa_gemm_sk_workspace()
// bytes whose first 4096 were zeroed once by the caller after allocation (epoch flags; never reset afterwards).
// rider (may be NULL): see SkRider; *rode <- 1 when it was put on this launch's idle CUs (>= 16 of them, bf16), else 0
// and the caller launches it on its own.
int fvqa_fewrows_takes(int M, int N, int K, int dtype, int out_dtype, int epilogue, size_t scratch_bytes);
int fvqa_fewrows_impl(const void* A, const void* B, void* C, const void* R, void* C2, float* scratch, int M, int N, int K, int lda,
                      int ldb, int ldc, int out_dtype, int epilogue, hipStream_t st);

int fvqa_gemm_sk_impl(const void* A, const void* B, void* C, const void* R, void* ws, size_t ws_bytes, int M, int N,
                      int K, int lda, int ldb, int ldc, int dtype, int out_dtype, int epilogue, hipStream_t st,
                      const fvqa_sk_rider* rider, int* rode, void* C2, const fvqa_sk_rope* rope) {
  if (rode) *rode = 0;
  if ((epilogue == FVQA_EPI_ROPE) != (rope != nullptr)) return FVQA_EINVAL;
  if (rope && (dtype != FVQA_H16 || out_dtype != FVQA_H16 || !rope->cos_t || !rope->sin_t || rope->seq_len <= 0 ||
               rope->head_dim <= 0 || (rope->head_dim % 8) || rope->cols < 0 || rope->cols > N || (rope->cols % rope->head_dim)))
    return FVQA_EINVAL;
  if (!ws || ws_bytes < fvqa_gemm_sk_workspace() || ((uintptr_t)ws & 255)) return FVQA_EALIGN;
  {
    // the launch epoch is a host-side kernel argument: a captured launch would be replayed with a stale epoch, its
    // partners' flags would already match and partial tiles would race — refuse instead of corrupting silently
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(st, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone) return FVQA_EINVAL;
  }
  // 17..64 rows against a 7B-class weight matrix (the tail rows' projections): a weight stream, its own kernels (gemm_fewrows.hip)
  if (!rider && !rope && fvqa_fewrows_takes(M, N, K, dtype, out_dtype, epilogue, ws_bytes - SYNC_BYTES)) {
    if ((epilogue == FVQA_EPI_SWIGLU_FWD_ST) && (!C2 || ((uintptr_t)C2 & 15))) return FVQA_EINVAL;
    return fvqa_fewrows_impl(A, B, C, R, C2, (float*)((char*)ws + SYNC_BYTES), M, N, K, lda, ldb, ldc, out_dtype, epilogue, st);
  }
  const int n_cu = cu_count();
  SkArgs a;
  a.A = A; a.B = B; a.C = C; a.R = R; a.C2 = C2;
  if ((epilogue == FVQA_EPI_SWIGLU_FWD || epilogue == FVQA_EPI_SWIGLU_FWD_ST) && (!C2 || (N & 31) || ((uintptr_t)C2 & 15) || out_dtype != dtype)) return FVQA_EINVAL;
  if (dtype == FVQA_H16) {
    // Outputs wide enough to fill the chip with whole tiles go to the 4-wave kernel (gemm4w.hip: one wave per SIMD, tile
    // width chosen per problem); a launch this kernel would split along K in full (N = 4096 outputs at M = 1024: 64 tiles
    // for 256 CUs) stays here.
    const int cus = n_cu < 256 ? n_cu : 256;
    const fvqa_sk_plan p0 = fvqa_sk_make_plan(M, N, K, 64, cus);
    // (the 4-wave loop addresses its operands with 32-bit DMA offsets: operands of 2 GiB or more — ~97k rows at K = 11008 —
    // stay with the 8-wave kernel below instead of failing; round-4 advisor finding)
    const bool dma32 = (size_t)M * lda * 2 < 0x7fffffffull && (size_t)N * ldb * 2 < 0x7fffffffull;
    if (dma32 && !(p0.full == 0 && p0.s > 1)) {
      const int nbt = fvqa_gemm4w_choose(M, N, K, dtype, out_dtype, epilogue, rider, cus);
      if (nbt) {
        unsigned long long* cst = nullptr;
        unsigned long long ep = 0;
#ifdef FVQA_SK_CLOCK
        ep = g_epoch.fetch_add(1) + 1;
        cst = (u64*)((char*)ws + SYNC_BYTES + (size_t)256 * SLAB_FLOATS * sizeof(float)) + (size_t)(ep % CLOCK_RING) * 256 * 16;
#endif
        return fvqa_gemm4w_impl(nbt, A, B, C, R, M, N, K, lda, ldb, ldc, out_dtype, epilogue, st, rider, rode, C2, rope, cus, cst,
                                ep);
      }
    }
  }
  a.sync = (u64*)ws;
  a.slabs = (float*)((char*)ws + SYNC_BYTES);
  a.stamps = (u64*)((char*)ws + SYNC_BYTES + (size_t)256 * SLAB_FLOATS * sizeof(float));
  a.M = M; a.N = N; a.K = K; a.lda = lda; a.ldb = ldb; a.ldc = ldc;
  a.plan = fvqa_sk_make_plan(M, N, K, dtype == FVQA_H16 ? 64 : 32, n_cu < 256 ? n_cu : 256);
  if (a.plan.n_teams * a.plan.ts > 256 || a.plan.n_teams * a.plan.ts > n_cu) return FVQA_ESHAPE;
  a.epoch = g_epoch.fetch_add(1) + 1;
#ifdef FVQA_SK_CLOCK
  a.stamps += (size_t)(a.epoch % CLOCK_RING) * 256 * 16;
#endif
  a.rider = SkRider{nullptr, nullptr, nullptr, 0, 0, 0, 0, 0, 0, 0, 0};
  a.rope_cos = rope ? rope->cos_t : nullptr; a.rope_sin = rope ? rope->sin_t : nullptr;
  a.rope_S = rope ? rope->seq_len : 1; a.rope_cols = rope ? rope->cols : 0; a.rope_hp = rope ? rope->head_dim / 2 : 1;
  const int idle = (n_cu < 256 ? n_cu : 256) - a.plan.n_teams * a.plan.ts;
  static const bool ride = !(getenv("FVQA_RIDER") && getenv("FVQA_RIDER")[0] == '0');   // tuning: FVQA_RIDER=0 keeps riders as own launches
  if (ride && rider && dtype == FVQA_H16 && idle >= 16 && rider->M >= 1 && rider->M <= 16 && (rider->K % 256) == 0 &&
      rider->N > 0 && rider->A && rider->B && rider->C) {
    const int strips = (rider->N + 15) / 16;
    a.rider = SkRider{rider->A, rider->B, rider->C, rider->M, rider->N, rider->K, rider->lda, rider->ldb, rider->ldc,
                      rider->accumulate_f32, idle < strips ? idle : strips};
    if (rode) *rode = 1;
  }
  // pure split launches of bf16 tiles with 24-bit slabs (2 or 4 pieces per tile: the N = 4096 outputs at M = 1024) run on the
  // 4-wave main loop (FVQA_GEMM4W_SK=0: the 8-wave kernel, for A/B runs)
  static const bool sk4 = !(getenv("FVQA_GEMM4W") && getenv("FVQA_GEMM4W")[0] == '0') &&
                          !(getenv("FVQA_GEMM4W_SK") && getenv("FVQA_GEMM4W_SK")[0] == '0');
  if (sk4 && SK_PACK24 && dtype == FVQA_H16 && out_dtype == FVQA_H16 && a.plan.full == 0 && a.plan.s >= 2 && a.plan.s <= 4 &&
      (epilogue == FVQA_EPI_NONE || epilogue == FVQA_EPI_RESIDUAL) && (size_t)M * lda * 2 < 0x7fffffffull &&
      (size_t)N * ldb * 2 < 0x7fffffffull) {
    a.rope_hmask = 0;
    return epilogue == FVQA_EPI_NONE ? launch_sk4<bf16_t, FVQA_EPI_NONE>(a, st) : launch_sk4<bf16_t, FVQA_EPI_RESIDUAL>(a, st);
  }
#define SK(T, TO)                                                                             \
  switch (epilogue) {                                                                         \
    case FVQA_EPI_NONE: return launch_sk<T, TO, FVQA_EPI_NONE>(a, st);                        \
    case FVQA_EPI_RESIDUAL: return launch_sk<T, TO, FVQA_EPI_RESIDUAL>(a, st);                \
    case FVQA_EPI_SWIGLU_BWD: return launch_sk<T, TO, FVQA_EPI_SWIGLU_BWD>(a, st);            \
    case FVQA_EPI_SWIGLU_FWD: return launch_sk<T, TO, FVQA_EPI_SWIGLU_FWD>(a, st);            \
    case FVQA_EPI_SWIGLU_BWD_ST: return launch_sk<T, TO, FVQA_EPI_SWIGLU_BWD_ST>(a, st);      \
    case FVQA_EPI_SWIGLU_FWD_ST: return launch_sk<T, TO, FVQA_EPI_SWIGLU_FWD_ST>(a, st);      \
    case FVQA_EPI_ROPE: return launch_sk<T, TO, FVQA_EPI_ROPE>(a, st);                        \
    default: return FVQA_EINVAL;                                                              \
  }
  if (dtype == FVQA_H16) {
    if (out_dtype == FVQA_F32) {
      if (epilogue != FVQA_EPI_NONE) return FVQA_EINVAL;
      return launch_sk<bf16_t, float, FVQA_EPI_NONE>(a, st);
    }
    SK(bf16_t, bf16_t)
  }
  SK(float, float)
#undef SK
}
