#!/usr/bin/env python3
"""Generates flipped-vqa_amd/csrc/gemm4w_asm.h: the main loop of the 4-wave projection GEMM (one wave per SIMD, gfx950) as ONE
inline-asm statement per tile width, with every register named here.

Why generated assembly: with one wave per SIMD a wave's accumulators (16*NBT registers per lane) live in the accumulator half
of the 512-entry register file; hipcc (ROCm 7.2) given the same loop as HIP source moves accumulators between the two halves
inside the loop (76-198 v_accvgpr_* per stage measured on the source form) and spills. Here the accumulators are bound to
a[16*j : 16*j+15] through physical-register constraints ("+{a[0:15]}"), so the compiler keeps them there across the statement
and C++ code after it reads them as ordinary values.

Loop structure (see gemm4w_loop.h for the design notes): per wide stage (128 B of K per row = two k-steps of
v_mfma_f32_16x16x32_bf16) two phases of 4*NBT MFMAs; fragment reads of the next k-step two per MFMA group in the first
(4+NBT)/2 groups of a phase; LDS-DMA (buffer_load_dwordx4 ... offen lds) of the weight pieces of stage u+2 spread over
phase 0 and of the activation pieces of stage u+2 over phase 1; one s_barrier in the middle of a stage behind a counted vmcnt.
A stage past the end of the K range is "loaded" through a descriptor with zero records: every lane is out of range, the
hardware writes zeros and fetches nothing, so every stage issues the same number of DMA instructions and all waits are counted.
A DMA piece costs the wave two instructions: its LDS destination into M0 (ahead of an MFMA, which is the required wait state)
and the load, whose per-lane offset (row block of the piece + swizzled chunk) sits in a register of its own and whose K offset
is the scalar offset of the stage.

Register map (per wave):
  a[16*j + 4*i : +3]   accumulator of output rows 16*i.. of this wave's 64, columns 16*j..   (i < 4, j < NBT)
  v[64:79]  set 0 activation fragments   v[80:80+4*NBT-1]   set 0 weight fragments
  v[144:159] set 1 activation fragments  v[160:160+4*NBT-1] set 1 weight fragments
  v[224:229] read addresses              v[232:239] / v[240:247] per-piece DMA offsets (A / B)      s[44:95] scalars
"""
import os
import sys

SA, SB = 2, 3
A_STAGE = 256 * 128


def gen(NBT):
    B_STAGE = 16 * NBT * 128
    PIECES_B = 2 * NBT
    NPB = (PIECES_B + 3) // 4
    NR = 4 + NBT
    RG = (NR + 1) // 2
    L = []
    e = L.append

    def A0(i): return f"v[{64 + 4 * i}:{67 + 4 * i}]"
    def B0(j): return f"v[{80 + 4 * j}:{83 + 4 * j}]"
    def A1(i): return f"v[{144 + 4 * i}:{147 + 4 * i}]"
    def B1(j): return f"v[{160 + 4 * j}:{163 + 4 * j}]"
    def ACC(i, j): return f"a[{16 * j + 4 * i}:{16 * j + 4 * i + 3}]"

    S_U, S_OOB, S_KOFF, S_SAOFF, S_SBOFF, S_SB1, S_SB2, S_T, S_T2, S_M0A, S_M0B, S_SA1 = (f"s{n}" for n in range(60, 72))
    PA = [f"s{72 + q}" for q in range(8)]
    PB = [f"s{80 + q}" for q in range(NPB)]
    PBL = [f"s{88 + q}" for q in range(NPB)]
    V_RA1, V_RB1, V_RA0N, V_RB0N, V_RDA1, V_RDB1 = (f"v{n}" for n in range(224, 230))
    VA = [f"v{232 + q}" for q in range(8)]                       # per-lane DMA offset of activation piece q
    VB = [f"v{240 + q}" for q in range(NPB)]                     # ... of this wave's q-th weight piece
    RS_A, RS_B = "s[44:47]", "s[48:51]"                          # descriptors of the stage being issued (all-OOB past the end)
    S_DA, S_DB = "s52", "s53"                                    # LDS destination bases (wave's A pieces / B ring) + slot

    def dma_a_m0(q):
        e(f"s_add_u32 m0, {S_DA}, {q * 1024}")

    def dma_a_ld(q):
        e(f"buffer_load_dwordx4 {VA[q]}, {RS_A}, {S_KOFF} offen lds")

    def dma_b_m0(q):
        e(f"s_add_u32 m0, {S_DB}, {PBL[q]}")

    def dma_b_ld(q):
        e(f"buffer_load_dwordx4 {VB[q]}, {RS_B}, {S_KOFF} offen lds")

    def dma_a(q):
        dma_a_m0(q); e("s_nop 0"); dma_a_ld(q)

    def dma_b(q):
        dma_b_m0(q); e("s_nop 0"); dma_b_ld(q)

    def select_rsrc(live_scc):
        """descriptors of the stage to issue: the operands' own, or (stage past the end: SCC = 0) with zero records — every
        lane out of range, the hardware writes zeros and fetches nothing"""
        assert live_scc
        e(f"s_cselect_b32 s46, s54, 0")
        e(f"s_cselect_b32 s50, s55, 0")

    def read(setn, r, va, vb):
        """fragment read number r of a k-step into set `setn`"""
        if r >= NR:
            return
        if r < 4:
            dst = (A0 if setn == 0 else A1)(r)
            e(f"ds_read_b128 {dst}, {va} offset:{r * 2048}")
        else:
            j = r - 4
            dst = (B0 if setn == 0 else B1)(j)
            e(f"ds_read_b128 {dst}, {vb} offset:{j * 2048}")

    # ---------------- prologue
    e(f"; ---- 4-wave ring loop, NBT = {NBT}")
    # packed scalar operands (an asm statement takes at most 30 operands): 64-bit pairs copied to fixed registers
    e(f"s_mov_b64 s[44:45], %[rsA01]")                            # descriptor words 0-1 (base address) of A and B
    e(f"s_mov_b64 s[48:49], %[rsB01]")
    e(f"s_mov_b64 s[54:55], %[rsAB2]")                            # s54 / s55 = records (bytes) of A / B
    e(f"s_mov_b64 s[56:57], %[str8]")                             # s56 / s57 = bytes per 8 rows of A / B
    e(f"s_mov_b64 s[58:59], %[wl]")                               # s58 = w | rx8 << 8, s59 = lds0
    e(f"s_mov_b32 s46, s54")
    e(f"s_mov_b32 s47, 0x00020000")
    e(f"s_mov_b32 s50, s55")
    e(f"s_mov_b32 s51, 0x00020000")
    S_W, S_RX8, S_LDS0, S_STRA8, S_STRB8 = "s42", "s43", "s59", "s56", "s57"
    e(f"s_and_b32 {S_W}, s58, 0xff")
    e(f"s_lshr_b32 {S_RX8}, s58, 8")
    e(f"s_lshl_b32 {S_T}, {S_W}, 13")
    e(f"s_add_u32 {S_M0A}, {S_LDS0}, {S_T}")                      # this wave's 8 pieces of an A stage
    e(f"s_add_u32 {S_M0B}, {S_LDS0}, {SA * A_STAGE}")
    for q in range(8):                                           # per-lane offset of piece q: voffA + (q ^ rx8) * strA8
        e(f"s_xor_b32 {S_T}, {S_RX8}, {q}")
        e(f"s_mul_i32 {S_T}, {S_T}, {S_STRA8}")
        e(f"v_add_u32 {VA[q]}, {S_T}, %[voffA]")
    e(f"s_mul_i32 {S_T2}, {S_W}, {NPB}")
    for q in range(NPB):                                         # p = min(w * NPB + q, PIECES_B - 1)
        e(f"s_add_u32 {S_T}, {S_T2}, {q}")
        e(f"s_min_u32 {S_T}, {S_T}, {PIECES_B - 1}")
        e(f"s_lshl_b32 {PBL[q]}, {S_T}, 10")
        e(f"s_mul_i32 {S_T}, {S_T}, {S_STRB8}")
        e(f"v_add_u32 {VB[q]}, {S_T}, %[voffB]")
    e(f"v_xor_b32 {V_RDA1}, 64, %[rdA]")
    e(f"v_xor_b32 {V_RDB1}, 64, %[rdB]")
    e(f"s_mov_b32 {S_SAOFF}, 0")
    e(f"s_mov_b32 {S_SBOFF}, 0")
    e(f"s_mov_b32 {S_KOFF}, %[kb0]")
    e(f"s_mov_b32 {S_DA}, {S_M0A}")
    e(f"s_mov_b32 {S_DB}, {S_M0B}")
    # B(0), A(0) into slots 0; B(1), A(1) into slots 1 (stage 1 may lie past the end: zero-record descriptors)
    for q in range(NPB):
        dma_b(q)
    for q in range(8):
        dma_a(q)
    e(f"s_cmp_gt_u32 %[nw], 1")
    select_rsrc(True)
    e(f"s_add_u32 {S_KOFF}, %[kb0], 128")
    e(f"s_add_u32 {S_DA}, {S_M0A}, {A_STAGE}")
    e(f"s_add_u32 {S_DB}, {S_M0B}, {B_STAGE}")
    for q in range(NPB):
        dma_b(q)
    for q in range(8):
        dma_a(q)
    e(f"s_waitcnt vmcnt({NPB + 8})")
    e("s_barrier")
    for r in range(NR):
        read(0, r, "%[rdA]", "%[rdB]")
    e(f"s_mov_b32 {S_U}, 0")
    # ---------------- one stage per iteration
    e("L_stage_%=:")
    # slots of stage u+1 / u+2, issue parameters of stage u+2
    e(f"s_xor_b32 {S_SA1}, {S_SAOFF}, {A_STAGE}")
    e(f"s_add_u32 {S_SB1}, {S_SBOFF}, {B_STAGE}")
    e(f"s_cmp_lt_u32 {S_SB1}, {SB * B_STAGE}")
    e(f"s_cselect_b32 {S_SB1}, {S_SB1}, 0")
    e(f"s_add_u32 {S_SB2}, {S_SB1}, {B_STAGE}")
    e(f"s_cmp_lt_u32 {S_SB2}, {SB * B_STAGE}")
    e(f"s_cselect_b32 {S_SB2}, {S_SB2}, 0")
    e(f"s_add_u32 {S_T}, {S_U}, 2")
    e(f"s_lshl_b32 {S_T2}, {S_T}, 7")
    e(f"s_add_u32 {S_KOFF}, %[kb0], {S_T2}")
    e(f"s_cmp_lt_u32 {S_T}, %[nw]")
    select_rsrc(True)
    e(f"s_add_u32 {S_DB}, {S_M0B}, {S_SB2}")                      # B(u+2) -> slot sb2 (phase 0)
    e(f"s_add_u32 {S_DA}, {S_M0A}, {S_SAOFF}")                    # A(u+2) -> slot sa  (phase 1)
    e(f"v_add_u32 {V_RA1}, {S_SAOFF}, {V_RDA1}")
    e(f"v_add_u32 {V_RB1}, {S_SBOFF}, {V_RDB1}")
    e(f"v_add_u32 {V_RA0N}, {S_SA1}, %[rdA]")
    e(f"v_add_u32 {V_RB0N}, {S_SB1}, %[rdB]")
    e("s_waitcnt lgkmcnt(0)")                                     # set 0 has arrived
    # phase 0
    for j in range(NBT):
        e(f"v_mfma_f32_16x16x32_bf16 {ACC(0, j)}, {B0(j)}, {A0(0)}, {ACC(0, j)}")
        if j < RG:
            read(1, 2 * j, V_RA1, V_RB1)
        e(f"v_mfma_f32_16x16x32_bf16 {ACC(1, j)}, {B0(j)}, {A0(1)}, {ACC(1, j)}")
        if j < RG:
            read(1, 2 * j + 1, V_RA1, V_RB1)
        pieces = list(range((j * NPB) // NBT, ((j + 1) * NPB) // NBT))
        e(f"v_mfma_f32_16x16x32_bf16 {ACC(2, j)}, {B0(j)}, {A0(2)}, {ACC(2, j)}")
        if pieces:
            dma_b_m0(pieces[0])
        e(f"v_mfma_f32_16x16x32_bf16 {ACC(3, j)}, {B0(j)}, {A0(3)}, {ACC(3, j)}")
        if pieces:
            dma_b_ld(pieces[0])
            for q in pieces[1:]:
                dma_b(q)
    # middle
    e("s_waitcnt lgkmcnt(0)")                                     # set 1 has arrived; this wave is done reading stage u
    e(f"s_waitcnt vmcnt({NPB})")                                  # stage u+1 has landed (B(u+2) may be in flight)
    e("s_barrier")
    # phase 1
    for j in range(NBT):
        e(f"v_mfma_f32_16x16x32_bf16 {ACC(0, j)}, {B1(j)}, {A1(0)}, {ACC(0, j)}")
        if j < RG:
            read(0, 2 * j, V_RA0N, V_RB0N)
        e(f"v_mfma_f32_16x16x32_bf16 {ACC(1, j)}, {B1(j)}, {A1(1)}, {ACC(1, j)}")
        if j < RG:
            read(0, 2 * j + 1, V_RA0N, V_RB0N)
        pieces = list(range((j * 8) // NBT, ((j + 1) * 8) // NBT))
        e(f"v_mfma_f32_16x16x32_bf16 {ACC(2, j)}, {B1(j)}, {A1(2)}, {ACC(2, j)}")
        if pieces:
            dma_a_m0(pieces[0])
        e(f"v_mfma_f32_16x16x32_bf16 {ACC(3, j)}, {B1(j)}, {A1(3)}, {ACC(3, j)}")
        if pieces:
            dma_a_ld(pieces[0])
            for q in pieces[1:]:
                dma_a(q)
    e(f"s_mov_b32 {S_SAOFF}, {S_SA1}")
    e(f"s_mov_b32 {S_SBOFF}, {S_SB1}")
    e(f"s_add_u32 {S_U}, {S_U}, 1")
    e(f"s_cmp_lt_u32 {S_U}, %[nw]")
    e("s_cbranch_scc1 L_stage_%=")
    e("s_waitcnt vmcnt(0) lgkmcnt(0)")                            # zero-fill DMA and unused reads of the stages past the end
    e("s_nop 15")                                                 # MFMA results -> any reader after the statement
    e("s_nop 15")

    body = "\n".join(f'      "{ln}\\n"' for ln in L)
    accs = ", ".join(f'"+{{a[{16 * j}:{16 * j + 15}]}}"(acc[{j}])' for j in range(NBT))
    clob = ['"memory"', '"scc"', '"vcc"']
    clob += [f'"v{n}"' for n in range(64, 248)]
    clob += [f'"s{n}"' for n in range(42, 96)]
    clob_s = ", ".join(clob)
    return f'''
template <> struct Ring4Asm<{NBT}> {{
  // rsA01 / rsB01: base addresses (descriptor words 0-1); rsAB2: records of A | records of B << 32; str8: bytes per 8 rows of
  // A | of B << 32; wl: (w | rx8 << 8) | lds0 << 32
  static __device__ __forceinline__ void run(f32x16 (&acc)[{NBT}], unsigned long long rsA01, unsigned long long rsB01,
                                             unsigned long long rsAB2, unsigned long long str8, unsigned long long wl,
                                             unsigned kb0, unsigned nw, unsigned voffA, unsigned voffB, unsigned rdA,
                                             unsigned rdB) {{
    asm volatile(
{body}
      : {accs}
      : [rsA01] "s"(rsA01), [rsB01] "s"(rsB01), [rsAB2] "s"(rsAB2), [str8] "s"(str8), [wl] "s"(wl), [kb0] "s"(kb0), [nw] "s"(nw),
        [voffA] "v"(voffA), [voffB] "v"(voffB), [rdA] "v"(rdA), [rdB] "v"(rdB)
      : {clob_s});
  }}
}};
'''


HEADER = '''// GENERATED by tools/gen_gemm4w_asm.py — do not edit; edit the generator and re-run it.
// Main loop of the 4-wave projection GEMM (gfx950, bf16, one wave per SIMD) as one inline-asm statement per tile width.
// Design notes: gemm4w_loop.h; register map and schedule: the generator's docstring.
#pragma once
#include "common.h"

namespace fvqa_ring4 {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int NBT> struct Ring4Asm;
'''

if __name__ == "__main__":
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = os.path.join(root, "flipped-vqa_amd", "csrc", "gemm4w_asm.h")
    widths = [int(x) for x in sys.argv[1:]] or [16, 12, 11]
    with open(out, "w") as f:
        f.write(HEADER)
        for n in widths:
            f.write(gen(n))
        f.write("\n}  // namespace fvqa_ring4\n")
    print(out)
