// 4-wave projection GEMM for gfx950 (bf16): one 256-thread workgroup per CU, ONE wave per SIMD, each wave 64 rows x the
// whole width of a 256 x (16*NBT) output tile with its accumulators in the accumulator half of the register file
// (main loop: gemm4w_loop.h / generated gemm4w_asm.h). Whole tiles only — no split along K, no exchange: it takes the
// projections whose outputs are wide enough to fill the chip with whole tiles (reference llama/model.py:89 q|k|v, :142
// w1|w3 and the dX of w2, :348 the LM head), and picks the tile WIDTH per problem (256 / 192 / 176 columns) so that the tile
// count lands on a multiple of the CU count: 12288 columns = 64 x 192 (256 tiles of 256 rows at M = 1024 instead of 192),
// 11008 = 57.3 x 192 (232 tiles instead of 172), 22016 = 114.7 x 192 (460 = 1.8 rounds instead of a whole round + a round of
// half tiles that exchange partial sums). The N = 4096 outputs (64 tiles for 256 CUs) stay with the split-K kernel of
// gemm_sk.hip, as does the fp32 build.
//
// Epilogues (applied to finished tiles staged through the wave's share of the idle ring memory so that every global
// access is a whole 16-byte chunk of a row): none (bf16 or fp32 out), + residual, RoPE of the q | k columns, SwiGLU forward
// with the backward's factors saved (W1 | W3 columns interleaved in blocks of 16, "AB16"), SwiGLU' (dH W2^T). Same
// arithmetic, element for element, as the epilogues of gemm_sk.hip (the two kernels produce bitwise-equal whole tiles).
//
// Rider: a second product of <= 16 rows (gemm_skinny.h) runs inside the launch on the workgroups that have no tile in the
// last round ("light" workgroups), spread over the XCDs by the same slot map that keeps the m tiles of a weight panel on
// one XCD.
#include "gemm4w_kernel.h"
#include "probe.h"
#include <atomic>
#include <cstdlib>

namespace {
using namespace fvqa_g4;

template <int NBT, typename TO, int EPI>
int launch4(const G4Args& a, hipStream_t st) {
  auto k = gemm4w_k<NBT, TO, EPI>;
  static std::atomic<unsigned long long> attr_done{0};          // one bit per device (fvqa_attr_needed)
  if (fvqa_attr_needed(attr_done)) {
    (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  }
  // dynamic LDS: the rings of the main loop; a launch that carries a rider asks for the rider's DMA ring + partial blocks if larger
  const int lds = a.rider.on && a.rider.dma && Geo<NBT>::RING_BYTES < FVQA_SKINNY_DMA_LDS ? FVQA_SKINNY_DMA_LDS : Geo<NBT>::RING_BYTES;
  {
    FvqaProbeScope ts(st, 2.0 * a.M * a.N * a.K, EPI | (sizeof(TO) == 4 ? 32 : 0) | 128);       // kind bit 7: the 4-wave kernel
    hipLaunchKernelGGL(k, dim3(a.grid), dim3(256), lds, st, a);
  }
  FVQA_CHECK_LAUNCH();
  return FVQA_OK;
}

template <int NBT>
int dispatch4(const G4Args& a, int out_dtype, int epilogue, hipStream_t st) {
  if (out_dtype == FVQA_F32) return launch4<NBT, float, FVQA_EPI_NONE>(a, st);
  switch (epilogue) {
    case FVQA_EPI_NONE: return launch4<NBT, bf16_t, FVQA_EPI_NONE>(a, st);
    case FVQA_EPI_RESIDUAL: return launch4<NBT, bf16_t, FVQA_EPI_RESIDUAL>(a, st);
    case FVQA_EPI_ROPE: return launch4<NBT, bf16_t, FVQA_EPI_ROPE>(a, st);
    case FVQA_EPI_SWIGLU_BWD_ST: return launch4<NBT, bf16_t, FVQA_EPI_SWIGLU_BWD_ST>(a, st);
    case FVQA_EPI_SWIGLU_FWD_ST:
      if constexpr (NBT % 2 == 0) return launch4<NBT, bf16_t, FVQA_EPI_SWIGLU_FWD_ST>(a, st);
      else return FVQA_EINVAL;
    default: return FVQA_EINVAL;
  }
}

}  // namespace

// ---- host side --------------------------------------------------------------------------------------------------------
// Estimated launch time (us) of an (M, N, K) problem with tiles of 16*nbt columns on n_cu CUs, fitted to the launch times of
// the C2 projections on an MI355X (profiles/r04_gemm4w.log): whole rounds of tiles; a round's time grows with the tile width
// and, sub-linearly, with the number of busy CUs (the chip holds a higher clock with fewer of them: 0.865 of the full-chip
// tile time with 192 busy, 0.82 with 172); an epilogue term per round; a rider streams its weight rows in whole passes of 32
// columns at ~17.5 GB/s per light workgroup from the start of the last round (re-fitted in round 5 against the width survey
// of tools/gemm4w_widths.py) and is a launch of its own (~3.2 TB/s + 10 us) when fewer than 8 workgroups are light.
// FVQA_RIDER_DMA=0 keeps the register form of the rider's strips (A/B runs); read once
static bool fvqa_rider_dma_enabled() {
  static const bool on = !(getenv("FVQA_RIDER_DMA") && getenv("FVQA_RIDER_DMA")[0] == '0');
  return on;
}

static double g4_cost_us(int M, int N, int K, int nbt, int epilogue, int out_dtype, const fvqa_sk_rider* rider, int n_cu,
                         int* light_out) {
  const int tm = (M + 255) / 256, tn = (N + 16 * nbt - 1) / (16 * nbt);
  const int tiles = tm * tn;
  const int rounds = (tiles + n_cu - 1) / n_cu;
  const int last = tiles - (rounds - 1) * n_cu;
  const int light = n_cu - last;
  const double t256 = 101.0 * K / 4096.0;
  // 11-block tiles cost 4 % more per column than the wider ones — and a quarter more in a multi-round SwiGLU' launch (measured,
  // profiles/r05_gemm4w_widths.log: 3072 x 11008: 331.9 us against 262.3 at 12 blocks where this model said 250 against 254)
  const double eff = nbt == 11 ? (epilogue == FVQA_EPI_SWIGLU_BWD_ST && rounds > 1 ? 1.25 : 1.04) : 1.0;
  double epi = out_dtype == FVQA_F32 ? 4.0 : 4.0;
  if (epilogue == FVQA_EPI_ROPE || epilogue == FVQA_EPI_RESIDUAL) epi = 9.0;
  if (epilogue == FVQA_EPI_SWIGLU_FWD_ST) epi = 7.0;
  if (epilogue == FVQA_EPI_SWIGLU_BWD_ST) epi = 17.0;
  auto round_us = [&](int busy) { return t256 * (nbt / 16.0) * (0.46 + 0.54 * busy / (double)n_cu) * eff + epi * (nbt / 16.0); };
  const double head = (rounds - 1) * round_us(n_cu);
  double total = head + round_us(last);
  if (rider) {
    const double bytes = 2.0 * rider->N * (double)rider->K;
    if (light >= 8) {
      // a light workgroup takes whole passes of two 16-column strips (32 x K weight elements) at ~17.5 GB/s: round 5's width
      // survey (profiles/r05_gemm4w_widths.log) — S = 384: 35 light workgroups x 4 passes of 512 KiB finished 120 us after the
      // first round, 30 us behind the tiles (the round-4 figure, 23 GB/s without the pass quantum, had it level with them and
      // picked that width: 11.5 % slower than the best one)
      // ... and with the operands streamed by LDS-DMA (gemm_skinny.h skinny_strip2_dma_4w: K ranges of whole 64-element stages)
      // ~49 GB/s after ~3.2 us of pipeline fill and partial-block sum per pass (fitted to two rider-bound launches of the width
      // survey: 24 light workgroups x 6 passes of 512 KiB end with the 232 tiles of dH W2^T at 12 column blocks, 84.6 us against
      // 82.7 without a rider — the register form needed 139.5; 24 x 14 passes of 320 KiB at 13B take 141 us)
      const int pairs = (rider->N + 31) / 32;
      const int passes = (pairs + light - 1) / light;
      const double pass_bytes = 2.0 * 32 * (double)rider->K;
      const bool dma = fvqa_rider_dma_enabled() && ((rider->K / 8) % 64) == 0;
      const double t_r = head + passes * (dma ? 3.2 + pass_bytes / 49.0e3 : pass_bytes / 17.5e3) + 3.0;
      const double main_us = total;
      if (t_r > total) total = t_r;
      total += 0.01 * main_us;                        // between rider-bound widths, the one whose tiles finish earlier
    } else {
      total += 10.0 + bytes / 3.2e6;
    }
  }
  if (light_out) *light_out = light;
  return total;
}

// Test / tuning override of the tile width (include/fvqa.h fvqa_gemm4w_force): per host thread, 0 = the cost model.
static thread_local int g_force_nbt = 0;
extern "C" int fvqa_gemm4w_force(int nbt) {
  if (nbt != 0 && nbt != 11 && nbt != 12 && nbt != 13 && nbt != 14 && nbt != 16) return FVQA_EINVAL;
  const int prev = g_force_nbt;
  g_force_nbt = nbt;
  return prev;
}

// Tile width (in 16-column blocks) the 4-wave kernel would use, 0 when the problem is not its to take.
extern "C" int fvqa_gemm4w_choose(int M, int N, int K, int dtype, int out_dtype, int epilogue, const fvqa_sk_rider* rider,
                                  int n_cu) {
  static const char* env = getenv("FVQA_GEMM4W");
  if (env && env[0] == '0') return 0;
  if (dtype != FVQA_H16 || (out_dtype != FVQA_H16 && out_dtype != FVQA_F32)) return 0;
  if (out_dtype == FVQA_F32 && epilogue != FVQA_EPI_NONE) return 0;
  if (epilogue != FVQA_EPI_NONE && epilogue != FVQA_EPI_RESIDUAL && epilogue != FVQA_EPI_ROPE &&
      epilogue != FVQA_EPI_SWIGLU_FWD_ST && epilogue != FVQA_EPI_SWIGLU_BWD_ST)
    return 0;
  // (a single ragged row tile over a large weight matrix — the LM head on the scored rows — is a weight stream: its too)
  const bool stream1 = M > 16 && M < 192 && (size_t)N * (size_t)K >= (size_t)100000000 && epilogue == FVQA_EPI_NONE;
  if ((M < 192 && !stream1) || N < 256 || (K % 64) || (N & 7) || n_cu < 8 || n_cu > 256) return 0;
  static const char* force = getenv("FVQA_GEMM4W_NBT");
  const int forced = g_force_nbt ? g_force_nbt : (force && force[0] ? atoi(force) : 0);
  int best = 0;
  double best_c = 1e30;
  for (int nbt : {16, 14, 13, 12, 11}) {
    if (forced && forced != nbt) continue;
    if (epilogue == FVQA_EPI_SWIGLU_FWD_ST && (nbt & 1)) continue;     // (a, b) pairs of 16-column blocks must not straddle tiles
    const double c = g4_cost_us(M, N, K, nbt, epilogue, out_dtype, rider, n_cu, nullptr);
    if (c < best_c) { best_c = c; best = nbt; }
  }
  return best;
}

// C[M,N] = A[M,K] x B[N,K]^T with the epilogue applied per finished tile (bf16 operands). The caller (fvqa_gemm_sk_impl)
// has validated pointers, alignment and leading dimensions. *rode <- 1 when the rider ran inside the launch.
int fvqa_gemm4w_impl(int nbt, const void* A, const void* B, void* C, const void* R, int M, int N, int K, int lda, int ldb,
                     int ldc, int out_dtype, int epilogue, hipStream_t st, const fvqa_sk_rider* rider, int* rode, void* C2,
                     const fvqa_sk_rope* rope, int n_cu, unsigned long long* clock_stamps, unsigned long long epoch) {
  if (rode) *rode = 0;
  if ((size_t)M * lda * 2 >= 0x7fffffffull || (size_t)N * ldb * 2 >= 0x7fffffffull) return FVQA_ESHAPE;   // 32-bit DMA offsets
  G4Args a;
  a.A = (const bf16_t*)A; a.B = (const bf16_t*)B; a.C = C; a.R = (const bf16_t*)R; a.C2 = C2;
  a.M = M; a.N = N; a.K = K; a.lda = lda; a.ldb = ldb; a.ldc = ldc;
  a.tm = (M + 255) / 256;
  a.tn = (N + 16 * nbt - 1) / (16 * nbt);
  a.tiles = a.tm * a.tn;
  a.rider = G4Rider{nullptr, nullptr, nullptr, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  a.rope_cos = rope ? rope->cos_t : nullptr; a.rope_sin = rope ? rope->sin_t : nullptr;
  a.rope_S = rope ? rope->seq_len : 1; a.rope_cols = rope ? rope->cols : 0; a.rope_hp = rope ? rope->head_dim / 2 : 1;
  a.rope_hmask = (rope && (rope->head_dim & (rope->head_dim - 1)) == 0) ? rope->head_dim - 1 : 0;
#ifdef FVQA_SK_CLOCK
  a.clock_stamps = clock_stamps; a.epoch = epoch;
#else
  (void)clock_stamps; (void)epoch;
#endif
  static const bool ride = !(getenv("FVQA_RIDER") && getenv("FVQA_RIDER")[0] == '0');
  const bool rider_ok = ride && rider && rider->M >= 1 && rider->M <= 16 && (rider->K % 256) == 0 && rider->N > 0 &&
                        rider->A && rider->B && rider->C;
  a.grid = a.tiles < n_cu ? a.tiles : n_cu;
  if (rider_ok) a.grid = n_cu;                                // light workgroups exist only on a full grid
  a.rounds = (a.tiles + a.grid - 1) / a.grid;
  if (rider_ok) {
    const int light = a.grid - (a.tiles - (a.rounds - 1) * a.grid);
    if (light >= 8) {
      a.rider = G4Rider{rider->A, rider->B, rider->C, rider->M, rider->N, rider->K, rider->lda, rider->ldb, rider->ldc,
                        rider->accumulate_f32, 1, fvqa_rider_dma_enabled() ? 1 : 0};
      if (rode) *rode = 1;
    } else {
      a.grid = a.tiles < n_cu ? a.tiles : n_cu;
      a.rounds = (a.tiles + a.grid - 1) / a.grid;
    }
  }
  switch (nbt) {
    case 16: return dispatch4<16>(a, out_dtype, epilogue, st);
    case 14: return dispatch4<14>(a, out_dtype, epilogue, st);
    case 13: return dispatch4<13>(a, out_dtype, epilogue, st);
    case 12: return dispatch4<12>(a, out_dtype, epilogue, st);
    case 11: return dispatch4<11>(a, out_dtype, epilogue, st);
    default: return FVQA_EINVAL;
  }
}
