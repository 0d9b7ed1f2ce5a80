// Main loop of the 4-wave projection GEMM for gfx950 (bf16): ONE wave per SIMD, each wave owns 64 rows x the whole tile
// width of a 256 x (16*NBT) output tile, so that a wave's accumulators (16*NBT registers per lane, in the accumulator half of
// the 512-entry register file) see 4 activation fragments and NBT weight fragments per MFMA k-step: (4 + NBT) LDS
// fragment reads per 4*NBT MFMAs — 0.31 per MFMA at NBT = 16 against 0.375 for the 8-wave loop of gemm256_loop.h, and the
// ratio stays good for the NARROW tiles (NBT = 12, 11) that let the N = 12288 / 11008 / 22016 projections of the step fill
// all 256 CUs with whole tiles (12288 = 64 x 192, 11008 = 62.5 x 176, 22016 = 125 x 176 or 62.5 x 352 columns).
//
// One call accumulates acc += A[m0.., k-range] x B[n0.., k-range]^T over `nw` wide stages (128 bytes of K per row = 64
// bf16 = two k-steps of v_mfma_f32_16x16x32_bf16 in NT form with the WEIGHT fragment as the row operand: a lane holds 4
// consecutive output columns of one row).
//   * LDS-DMA rings: activations A 2 x 32 KiB, weights B 3 x (2*NBT) KiB; every piece is 8 rows x 128 B = whole cache
//     lines, moved by `buffer_load_dwordx4 ... offen lds` with ONE per-lane offset register PER PIECE (row block of the piece
//     + swizzled chunk, computed once before the loop: tools/gen_gemm4w_asm.py VA / VB) and only the stage's K offset in the
//     SCALAR offset, so a DMA costs the wave no vector-ALU work inside the loop. The hardware range check covers the per-lane
//     offset only — which is why the ROW part of every address must stay there: rows past M / N then read as zeros, no
//     clamped pointers (moving the row block into the scalar offset would silently drop the zero-fill of edge tiles);
//   * a wave moves exactly the 64 activation rows it reads itself, plus its quarter of the weight rows;
//   * the image is lane-linear (DMA constraint), XOR-swizzled on the SOURCE offset and on the ds_read_b128 address
//     (chunk c of row r sits at chunk c ^ (r & 7): conflict-free fragment reads, as in gemm256_loop.h);
//   * per stage TWO phases of 4*NBT MFMAs: phase 0 runs k-step 0 while the fragments of k-step 1 are read and the
//     weight pieces of stage u+2 are issued; ONE s_barrier in the middle (stage u+1 has landed: counted vmcnt; every
//     wave has finished reading stage u); phase 1 runs k-step 1 while the k-step-0 fragments of stage u+1 are read and
//     the activation pieces of stage u+2 are issued. Fragment reads are issued in the first 10 MFMA groups of a phase and
//     waited for a whole phase later, so with one wave per SIMD nothing but a DMA issue ever sits between two MFMAs.
// The loop itself is ONE generated inline-asm statement per tile width (gemm4w_asm.h, tools/gen_gemm4w_asm.py): with the
// accumulators bound to a[0 : 16*NBT) by physical-register constraints; written as HIP source the same loop made hipcc move
// accumulators between the two register-file halves inside the loop (76-198 v_accvgpr_* per stage) and spill.
// On return every DMA of the call has landed and all of this wave's reads are done; the caller must barrier before the
// ring memory is reused.
#pragma once
#include "common.h"
#include "gemm4w_asm.h"

namespace fvqa_ring4 {

constexpr int TM = 256;
constexpr int A_STAGE = TM * 128;                       // 32 KiB
constexpr int SA = 2, SB = 3;                           // (the generator has the same constants)

template <int NBT> struct Geo {
  static constexpr int TN = 16 * NBT;
  static constexpr int B_STAGE = TN * 128;              // 2*NBT KiB
  static constexpr int RING_BYTES = SA * A_STAGE + SB * B_STAGE;
};

// acc[j][4*i + e]: output row 16*(i ^ rowxor/16) + (lane & 15) of this wave's 64, column 16*j + 4*(lane >> 4) + e.
// rowxor (0, 16, 32 or 48): register row block i of a wave holds tile row block i ^ (rowxor / 16) of its 64 rows.
// A, B: row-major with K contiguous, M x lda and N x ldb elements, both below 2 GiB (32-bit DMA offsets; rows past M / N
// are read as zeros by the buffer range check, which covers the per-lane offset only: the row part of every address is
// in the per-lane offset, the K part — always inside a row — in the scalar one).
template <int NBT>
__device__ __forceinline__ void ring4_loop(f32x16 (&acc)[NBT], unsigned lds0, const bf16_t* __restrict__ A,
                                           const bf16_t* __restrict__ B, int M, int N, int lda, int ldb, int m0, int n0,
                                           size_t kel0, int nw, int w, int lane, int rowxor = 0) {
  if (nw <= 0) return;
  const unsigned long long pa = (unsigned long long)(uintptr_t)A & 0xFFFFFFFFFFFFull, pb = (unsigned long long)(uintptr_t)B & 0xFFFFFFFFFFFFull;
  const unsigned long long rsAB2 = (unsigned long long)((unsigned)M * (unsigned)lda * 2u) | ((unsigned long long)((unsigned)N * (unsigned)ldb * 2u) << 32);
  const unsigned long long str8 = (unsigned long long)(8u * (unsigned)lda * 2u) | ((unsigned long long)(8u * (unsigned)ldb * 2u) << 32);
  const unsigned long long wl = (unsigned long long)((unsigned)w | (((unsigned)rowxor >> 3) << 8)) | ((unsigned long long)lds0 << 32);
  const int lr = lane >> 3, lc = (lane & 7) ^ lr;                       // row within a piece; swizzled 16-byte chunk
  const unsigned voffA = (unsigned)(m0 + 64 * w + lr) * (unsigned)lda * 2u + (unsigned)lc * 16u;
  const unsigned voffB = (unsigned)(n0 + lr) * (unsigned)ldb * 2u + (unsigned)lc * 16u;
  const int frow = lane & 15, fkc = lane >> 4, fsw = lane & 7;
  const unsigned ck0 = (unsigned)((fkc ^ fsw) << 4);
  const unsigned rdA = lds0 + (unsigned)((64 * w + frow) * 128) + ck0;
  const unsigned rdB = lds0 + (unsigned)(SA * A_STAGE) + (unsigned)(frow * 128) + ck0;
  Ring4Asm<NBT>::run(acc, pa, pb, rsAB2, str8, wl, (unsigned)(kel0 * 2), (unsigned)nw, voffA, voffB, rdA, rdB);
}

// The split-K form (256-column tiles, nw > 0): the loop and, inside the same statement, the exchange of the tile's pieces
// (Ring4AsmSK). xp: lane k holds exchange parameter k (Ring4AsmSK::XP_*); fa: lane q < np - 1 holds the address of partner
// q's epoch flag. On return register row blocks [0, 4 / np) hold the reduced tile rows of this piece, every wave has passed
// a barrier after its last ring read, and the ring memory may be reused.
__device__ __forceinline__ void ring4_loop_sk(f32x16 (&acc)[16], unsigned lds0, const bf16_t* __restrict__ A,
                                              const bf16_t* __restrict__ B, int M, int N, int lda, int ldb, int m0, int n0,
                                              size_t kel0, int nw, int w, int lane, int rowxor, unsigned xp,
                                              unsigned long long fa) {
  if (nw <= 0) return;                                    // (never taken: a piece has at least one stage)
  const unsigned long long pa = (unsigned long long)(uintptr_t)A & 0xFFFFFFFFFFFFull, pb = (unsigned long long)(uintptr_t)B & 0xFFFFFFFFFFFFull;
  const unsigned long long rsAB2 = (unsigned long long)((unsigned)M * (unsigned)lda * 2u) | ((unsigned long long)((unsigned)N * (unsigned)ldb * 2u) << 32);
  const unsigned long long str8 = (unsigned long long)(8u * (unsigned)lda * 2u) | ((unsigned long long)(8u * (unsigned)ldb * 2u) << 32);
  const unsigned long long wl = (unsigned long long)((unsigned)w | (((unsigned)rowxor >> 3) << 8)) | ((unsigned long long)lds0 << 32);
  const int lr = lane >> 3, lc = (lane & 7) ^ lr;
  const unsigned voffA = (unsigned)(m0 + 64 * w + lr) * (unsigned)lda * 2u + (unsigned)lc * 16u;
  const unsigned voffB = (unsigned)(n0 + lr) * (unsigned)ldb * 2u + (unsigned)lc * 16u;
  const int frow = lane & 15, fkc = lane >> 4, fsw = lane & 7;
  const unsigned ck0 = (unsigned)((fkc ^ fsw) << 4);
  const unsigned rdA = lds0 + (unsigned)((64 * w + frow) * 128) + ck0;
  const unsigned rdB = lds0 + (unsigned)(SA * A_STAGE) + (unsigned)(frow * 128) + ck0;
  Ring4AsmSK::run(acc, pa, pb, rsAB2, str8, wl, (unsigned)(kel0 * 2), (unsigned)nw, voffA, voffB, rdA, rdB, xp, fa);
}

}  // namespace fvqa_ring4
