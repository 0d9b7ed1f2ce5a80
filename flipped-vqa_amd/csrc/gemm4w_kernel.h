// Device side of the 4-wave projection GEMM (gemm4w.hip; also built into tools/gemm4w_probe.hip with in-kernel phase stamps).
#pragma once
#include "gemm4w_loop.h"
#include "gemm_skinny.h"

namespace fvqa_g4 {
using namespace fvqa_ring4;

struct G4Rider {
  const void* A; const void* B; void* C;
  int M, N, K, lda, ldb, ldc;
  int acc;               // 0: C (bf16) = product; 1: C (fp32) += product
  int on;                // 1: run it on the light workgroups of this launch
  int dma;               // 1: stream the weight rows by LDS-DMA (FVQA_RIDER_DMA=0 keeps the register form: A/B runs)
};

struct G4Args {
  const bf16_t* A; const bf16_t* B; void* C; const bf16_t* R;
  void* C2;              // FVQA_EPI_SWIGLU_FWD_ST: z (M, N/2), row stride N/2
  int M, N, K, lda, ldb, ldc;
  int tm, tn, tiles, grid, rounds;
  G4Rider rider;
  const float* rope_cos; const float* rope_sin;
  int rope_S, rope_cols, rope_hp, rope_hmask;     // hmask = head_dim - 1 when head_dim is a power of two, else 0
#ifdef FVQA_G4_STAMPS
  unsigned long long* stamps;      // tuning builds only: 8 x 100 MHz timestamps per workgroup, memory nothing else reads
#endif
#ifdef FVQA_SK_CLOCK
  unsigned long long* clock_stamps;   // diagnostic build (tools/sk_clock.py): this launch's slice of the stamp ring, or NULL
  unsigned long long epoch;
#endif
};

#ifdef FVQA_G4_STAMPS
#define G4_STAMP(slot) do { if (threadIdx.x == 0 && a.stamps) a.stamps[(size_t)blockIdx.x * 8 + (slot)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define G4_STAMP(slot) do { } while (0)
#endif

__device__ __forceinline__ int xcd_chunk(int bid, int nwg) {     // consecutive work ids share an XCD (bijective)
  const int q = nwg >> 3, r = nwg & 7, x = bid & 7;
  const int base = (x < r) ? x * (q + 1) : r * (q + 1) + (x - r) * q;
  return base + (bid >> 3);
}

// Slot (workgroup after xcd_chunk) -> position k in a round's tile list. Groups of `tm` consecutive positions (the m tiles
// of one weight panel) go to ONE XCD, consecutive groups to consecutive XCDs, so that the positions past the end of the
// last round — the light workgroups, and with them the idle share of a partly filled round — are spread over all XCDs.
// An XCD's share of `per` slots holds q = per / tm whole groups; the r = per - q tm slots left over take the positions behind
// all whole groups, r consecutive ones per XCD (round 5: with tm = 6 or 9 — 13B at 1536 rows, S = 384 — the map used to be the
// identity, which put every light workgroup and the whole idle share of the last round on the last one or two XCDs).
// Bijective on [0, grid) for every tm; identity when grid is not a multiple of 8 or tm exceeds the share.
__device__ __forceinline__ int slot_to_pos(int slot, int grid, int tm) {
  const int per = grid >> 3;
  if (grid & 7) return slot;
  const int q = per / tm, r = per - q * tm;
  const int x = slot / per, p = slot - x * per;
  if (p < q * tm) {
    const int gi = p / tm;
    return (gi * 8 + x) * tm + (p - gi * tm);
  }
  return 8 * q * tm + x * r + (p - q * tm);
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
__device__ __forceinline__ size_t ab16(int c) { return (size_t)(c >> 4) * 32 + (c & 15); }

// Finished tile -> memory. The wave's 64 x TN accumulators go through its private share of the (idle) ring in two passes
// of 32 rows (rows padded by one 16-byte chunk: conflict-free fragment writes), then every lane takes whole 8-column
// chunks of a row: 16-byte global accesses, consecutive lanes on consecutive chunks. The epilogue's global operands
// (residual rows, the saved SwiGLU factors, the RoPE table rows) of a pass are requested BEFORE its staging, so that their
// latency runs under the LDS round trip instead of once per chunk.
// n_own (1, 2 or 4) / rowxor: a split-K piece stores only the register row blocks [0, n_own) it has reduced; register row block
// i holds tile row block i ^ (rowxor / 16) of the wave's 64 rows (gemm_sk.hip). ARGS: G4Args or the split-K kernel's SkArgs.
template <int NBT, typename TO, int EPI, typename ARGS>
__device__ __forceinline__ void store_tile4(f32x16 (&acc)[NBT], char* smem, const ARGS& a, int m0, int n0, int w, int lane,
                                            int n_own = 4, int rowxor = 0) {
  constexpr int TN = 16 * NBT, RS = TN + 4;
  constexpr bool SWF = EPI == FVQA_EPI_SWIGLU_FWD_ST, SWB = EPI == FVQA_EPI_SWIGLU_BWD_ST;
  constexpr int IPR = SWF ? TN / 16 : TN / 8;               // work items per row
  constexpr int TOTAL = 32 * IPR;
  constexpr int IPL = (TOTAL + 63) / 64;                    // items per lane and pass
  constexpr bool PRE = SWB || EPI == FVQA_EPI_RESIDUAL || EPI == FVQA_EPI_ROPE;
  float* stg = reinterpret_cast<float*>(smem) + w * (32 * RS);
  const int frow = lane & 15, q4 = lane >> 4;
  const int M = a.M, N = a.N, ldc = a.ldc;
  const bf16_t* const Rp = reinterpret_cast<const bf16_t*>(a.R);
#pragma unroll
  for (int p = 0; p < 2; ++p) {
    if (n_own <= 2 * p) continue;                                         // (wave-uniform)
    // staged row r of the pass = register row block 2p + (r >> 4), i.e. tile row ((32p + (r & 16)) ^ rowxor) + (r & 15)
    const int mw0 = m0 + 64 * w;
    auto row_m = [&](int r) { return mw0 + ((32 * p + (r & 16)) ^ rowxor) + (r & 15); };
    auto live = [&](int r) { return 2 * p + (r >> 4) < n_own; };
    uint4 pq0[PRE ? IPL : 1], pq1[PRE ? IPL : 1];
    int pos0 = 0;
    if constexpr (EPI == FVQA_EPI_ROPE) pos0 = (mw0 + 32 * p) % a.rope_S;        // (wave-uniform; rows are in order when rowxor == 0)
    if constexpr (PRE) {
#pragma unroll
      for (int t = 0; t < IPL; ++t) {
        const int g = t * 64 + lane;
        const int r = g / IPR, it = g - r * IPR;
        const int m = row_m(r), n = n0 + 8 * it;
        pq0[t] = pq1[t] = uint4{0u, 0u, 0u, 0u};
        if ((TOTAL % 64 == 0 || g < TOTAL) && live(r) && m < M && n < N) {
          if constexpr (SWB) {
            const bf16_t* rp = Rp + (size_t)m * ldc + ab16(n);           // (s, t) rows: AB16, 2N columns
            pq0[t] = *reinterpret_cast<const uint4*>(rp);
            pq1[t] = *reinterpret_cast<const uint4*>(rp + 16);
          } else if constexpr (EPI == FVQA_EPI_RESIDUAL) {
            pq0[t] = *reinterpret_cast<const uint4*>(Rp + (size_t)m * ldc + n);
          } else {                                                         // RoPE: 4 cos + 4 sin of this chunk's pairs
            if (n < a.rope_cols) {
              int pos = pos0 + r;
              pos = (rowxor == 0 && a.rope_S >= 32) ? (pos >= a.rope_S ? pos - a.rope_S : pos) : m % a.rope_S;
              const int hi = a.rope_hmask ? (n & a.rope_hmask) : (n % (2 * a.rope_hp));
              const size_t ti = (size_t)pos * a.rope_hp + (hi >> 1);
              pq0[t] = *reinterpret_cast<const uint4*>(a.rope_cos + ti);
              pq1[t] = *reinterpret_cast<const uint4*>(a.rope_sin + ti);
            }
          }
        }
      }
    }
#pragma unroll
    for (int ii = 0; ii < 2; ++ii)
#pragma unroll
      for (int j = 0; j < NBT; ++j) {
        const int i = 2 * p + ii;
        *reinterpret_cast<f32x4*>(stg + (16 * ii + frow) * RS + 16 * j + 4 * q4) =
            f32x4{acc[j][4 * i], acc[j][4 * i + 1], acc[j][4 * i + 2], acc[j][4 * i + 3]};
      }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int t = 0; t < IPL; ++t) {
      const int g = t * 64 + lane;
      if (TOTAL % 64 && g >= TOTAL) continue;
      const int r = g / IPR, it = g - r * IPR;
      const int m = row_m(r);
      if (!live(r)) continue;
      if constexpr (SWF) {
        // the tile holds the W1 | W3 projections in AB16 order: blocks (2k, 2k+1) = (a, b) of 16 hidden units; this item is
        // half h of pair k. z = silu(a) * b from the values ROUNDED to bf16 (llama/model.py:142); the a / b slots of C get
        // the backward's factors s = silu(a) and t = dz/da = b sigma(a) (1 + a (1 - sigma(a)))  (FVQA_EPI_SWIGLU_FWD_ST)
        const int k = it >> 1, h = it & 1;
        const float* sp = stg + r * RS + 32 * k + 8 * h;
        const f32x4 a0 = *reinterpret_cast<const f32x4*>(sp), a1 = *reinterpret_cast<const f32x4*>(sp + 4);
        const f32x4 b0 = *reinterpret_cast<const f32x4*>(sp + 16), b1 = *reinterpret_cast<const f32x4*>(sp + 20);
        const float av_[8] = {a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
        const float bv_[8] = {b0[0], b0[1], b0[2], b0[3], b1[0], b1[1], b1[2], b1[3]};
        const int n = n0 + 32 * k + 8 * h;
        if (m >= M || n >= N) continue;
        float s_[8], t_[8], z_[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float av = round_to<bf16_t>(av_[e]), bv = round_to<bf16_t>(bv_[e]);
          const float sg = __builtin_amdgcn_rcpf(1.f + __expf(-av));
          const float sv = round_to<bf16_t>(av * sg);
          z_[e] = sv * bv;
          s_[e] = sv;
          t_[e] = bv * sg * (1.f + av * (1.f - sg));
        }
        bf16_t* cp = (bf16_t*)a.C + (size_t)m * ldc + n;
        *reinterpret_cast<uint4*>(cp) = pack8(s_);
        *reinterpret_cast<uint4*>(cp + 16) = pack8(t_);
        *reinterpret_cast<uint4*>((bf16_t*)a.C2 + (size_t)m * (N >> 1) + (n0 >> 1) + 16 * k + 8 * h) = pack8(z_);
      } else {
        const float* sp = stg + r * RS + 8 * it;
        const f32x4 lo = *reinterpret_cast<const f32x4*>(sp), hi = *reinterpret_cast<const f32x4*>(sp + 4);
        float v[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        const int n = n0 + 8 * it;
        if (m >= M || n >= N) continue;
        if constexpr (SWB) {
          // v = dz[m][n..]; R = (s, t) saved by the forward in the a / b slots of the AB16 rows (2N columns), C = d(a | b):
          // da = dz * t, db = dz * s   (llama/model.py:142 backward)
          const size_t o = (size_t)m * ldc + ab16(n);
          float s_[8], t_[8], da[8], db[8];
          unpack8(pq0[t], s_);
          unpack8(pq1[t], t_);
#pragma unroll
          for (int e = 0; e < 8; ++e) { da[e] = v[e] * t_[e]; db[e] = v[e] * s_[e]; }
          *reinterpret_cast<uint4*>((bf16_t*)a.C + o) = pack8(da);
          *reinterpret_cast<uint4*>((bf16_t*)a.C + o + 16) = pack8(db);
        } else {
          if constexpr (EPI == FVQA_EPI_ROPE) {
            // RoPE of the q | k columns where they are produced (model.py:61-67 on wq(x), wk(x)): 4 rotation pairs per chunk;
            // value rounded to bf16, rotated in fp32, rounded again at the store (the arithmetic of rope_qk_k)
            if (n < a.rope_cols) {
              const float cc[4] = {__uint_as_float(pq0[t].x), __uint_as_float(pq0[t].y), __uint_as_float(pq0[t].z), __uint_as_float(pq0[t].w)};
              const float ss[4] = {__uint_as_float(pq1[t].x), __uint_as_float(pq1[t].y), __uint_as_float(pq1[t].z), __uint_as_float(pq1[t].w)};
#pragma unroll
              for (int pr = 0; pr < 4; ++pr) {
                const float e = round_to<bf16_t>(v[2 * pr]), d = round_to<bf16_t>(v[2 * pr + 1]);
                v[2 * pr] = e * cc[pr] - d * ss[pr];
                v[2 * pr + 1] = e * ss[pr] + d * cc[pr];
              }
            }
          }
          if constexpr (EPI == FVQA_EPI_RESIDUAL) {
            float r_[8];
            unpack8(pq0[t], r_);
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] += r_[e];
          }
          if constexpr (sizeof(TO) == 2) {
            *reinterpret_cast<uint4*>((bf16_t*)a.C + (size_t)m * ldc + n) = pack8(v);
          } else {
            float* cp = (float*)a.C + (size_t)m * ldc + n;
            *reinterpret_cast<float4*>(cp) = make_float4(v[0], v[1], v[2], v[3]);
            *reinterpret_cast<float4*>(cp + 4) = make_float4(v[4], v[5], v[6], v[7]);
          }
        }
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }
}

template <int NBT, typename TO, int EPI>
__global__ __launch_bounds__(256) void gemm4w_k(const G4Args a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
  const int slot = xcd_chunk(blockIdx.x, a.grid);
  const int pos = slot_to_pos(slot, a.grid, a.tm);
  G4_STAMP(0);
  for (int r = 0; r < a.rounds; ++r) {
    const int t = r * a.grid + pos;
    if (t >= a.tiles) break;
    const int nt = t / a.tm, mt = t - nt * a.tm;             // consecutive positions = the m tiles of one weight panel
    const int m0 = mt * TM, n0 = nt * Geo<NBT>::TN;
    f32x16 acc[NBT];
#pragma unroll
    for (int j = 0; j < NBT; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;
#ifdef FVQA_SK_CLOCK
    // in-loop shader clock (MI355X_MICROARCH.md "DVFS give-back" item 6): s_memtime / s_memrealtime around the loop of the
    // first tile, same record as gemm_sk.hip's; the shipping build compiles none of it
    unsigned long long ct0, cr0, ct1, cr1;
    asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(ct0), "=s"(cr0) :: "memory");
#endif
    ring4_loop<NBT>(acc, lds0, a.A, a.B, a.M, a.N, a.lda, a.ldb, m0, n0, 0, a.K / 64, w, lane, 0);
#ifdef FVQA_SK_CLOCK
    asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(ct1), "=s"(cr1) :: "memory");
    if (tid == 0 && r == 0 && a.clock_stamps) {
      unsigned long long* st_ = a.clock_stamps + (size_t)slot * 16;
      st_[0] = ct0; st_[1] = cr0; st_[2] = ct1; st_[3] = cr1;
      st_[4] = ((unsigned long long)(unsigned)a.N << 32) | (unsigned)a.K;
      st_[5] = a.epoch;
      st_[6] = (unsigned long long)EPI | (1ull << 8) | ((unsigned long long)(a.K / 64) << 16) | ((unsigned long long)NBT << 32) |
               ((unsigned long long)sizeof(TO) << 40) | ((unsigned long long)a.grid << 48);
      st_[7] = (unsigned long long)a.M;
    }
#endif
    // Everything below is addressed from opaque copies of (lane, wave, tile origin): the epilogue's per-lane address
    // arithmetic must not be hoisted above the loop statement (it owns v64-v231 there)
    int lane_e = lane, w_e = w, m0_e = m0, n0_e = n0;
    asm volatile("" : "+v"(lane_e), "+s"(w_e), "+s"(m0_e), "+s"(n0_e));
    __syncthreads();                                         // every wave is done reading the ring
    G4_STAMP(1 + 2 * (r < 2 ? r : 1));
    store_tile4<NBT, TO, EPI>(acc, smem, a, m0_e, n0_e, w_e, lane_e);
    __syncthreads();                                         // staging reads done before the next tile's DMA
    G4_STAMP(2 + 2 * (r < 2 ? r : 1));
  }
  if (a.rider.on) {
    // light workgroups: positions of the last round past the end of the tile list
    const int krem = a.tiles - (a.rounds - 1) * a.grid;      // tiles in the last round (1 .. grid)
    const int light = a.grid - krem;
    if (pos >= krem) {
      const G4Rider& rd = a.rider;
      float(*part)[8][16][20] = reinterpret_cast<float(*)[8][16][20]>(smem);
      // operand rows streamed by LDS-DMA through the idle ring memory (gemm_skinny.h: the launch asks for FVQA_SKINNY_DMA_LDS
      // bytes when it carries a rider) when every K range is whole 64-element stages; same results either way
      const bool dma = rd.dma && ((rd.K / 8) % 64) == 0;
      char* ring = smem;
      for (int pair = pos - krem; pair * 32 < rd.N; pair += light) {     // two adjacent 16-column strips per pass
        if (dma) {
          if (rd.acc)
            skinny_strip2_dma_4w<float, FVQA_EPI_SKINNY_ACC>((const bf16_t*)rd.A, (const bf16_t*)rd.B, (float*)rd.C, rd.M, rd.N,
                                                             rd.K, rd.lda, rd.ldb, rd.ldc, pair * 32, ring);
          else
            skinny_strip2_dma_4w<bf16_t, FVQA_EPI_NONE>((const bf16_t*)rd.A, (const bf16_t*)rd.B, (bf16_t*)rd.C, rd.M, rd.N, rd.K,
                                                        rd.lda, rd.ldb, rd.ldc, pair * 32, ring);
        } else if (rd.acc)
          skinny_strip2_4w<float, FVQA_EPI_SKINNY_ACC>((const bf16_t*)rd.A, (const bf16_t*)rd.B, (float*)rd.C, rd.M, rd.N, rd.K,
                                                       rd.lda, rd.ldb, rd.ldc, pair * 32, part);
        else
          skinny_strip2_4w<bf16_t, FVQA_EPI_NONE>((const bf16_t*)rd.A, (const bf16_t*)rd.B, (bf16_t*)rd.C, rd.M, rd.N, rd.K,
                                                  rd.lda, rd.ldb, rd.ldc, pair * 32, part);
        __syncthreads();                                     // `part` is rewritten by the next pass
      }
      G4_STAMP(5);
    }
  }
}


}  // namespace fvqa_g4
