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

The split-K form (`Ring4AsmSK`, NBT = 16 only; gemm_sk.hip gemm4w_sk_k) appends the whole hand-off between the pieces of a tile to
the SAME statement ("split-K exchange" section below): its parameters arrive in one VGPR operand, parameter k in lane k
(v_readlane), the partners' flag addresses in a second one; s[96:99] and the registers the loop no longer needs carry the
diagnostic stamps (record address 0 in the shipping build = two scalar instructions and a branch per stamp).
"""
import os
import sys

SA, SB = 2, 3
A_STAGE = 256 * 128


def gen(NBT, sk=False):
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
    if sk:
        # diagnostic stamps (tools/sk_clock.py): xp lanes STAMP, STAMP + 1 = address of this workgroup's record, 0 = none (the
        # shipping build): s_memtime / s_memrealtime on both sides of the loop, stored by lane 0 of wave 0
        e(f"v_readlane_b32 s96, %[xp], {XP['STAMP']}")
        e(f"v_readlane_b32 s97, %[xp], {XP['STAMP'] + 1}")
        e("s_nop 4")
        e("s_cmp_eq_u64 s[96:97], 0")
        e("s_cbranch_scc1 L_nostamp0_%=")
        e(f"s_cmp_lg_u32 {S_W}, 0")
        e("s_cbranch_scc1 L_nostamp0_%=")
        e("s_memtime s[98:99]")
        e("s_waitcnt lgkmcnt(0)")
        e("v_mov_b32 v226, s98")
        e("v_mov_b32 v227, s99")
        e("s_memrealtime s[98:99]")
        e("s_waitcnt lgkmcnt(0)")
        e("v_mov_b32 v228, s98")
        e("v_mov_b32 v229, s99")
        e("v_mov_b32 v224, s96")
        e("v_mov_b32 v225, s97")
        e("s_mov_b64 s[98:99], exec")
        e("s_mov_b64 exec, 1")
        e("global_store_dwordx4 v[224:225], v[226:229], off")
        e("s_mov_b64 exec, s[98:99]")
        e("L_nostamp0_%=:")
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
    if sk:
        assert NBT == 16
        e("s_cmp_eq_u64 s[96:97], 0")
        e("s_cbranch_scc1 L_nostamp1_%=")
        e(f"s_cmp_lg_u32 {S_W}, 0")
        e("s_cbranch_scc1 L_nostamp1_%=")
        e("s_memtime s[98:99]")
        e("s_waitcnt lgkmcnt(0)")
        e("v_mov_b32 v226, s98")
        e("v_mov_b32 v227, s99")
        e("s_memrealtime s[98:99]")
        e("s_waitcnt lgkmcnt(0)")
        e("v_mov_b32 v228, s98")
        e("v_mov_b32 v229, s99")
        e("v_mov_b32 v224, s96")
        e("v_mov_b32 v225, s97")
        e("s_mov_b64 s[98:99], exec")
        e("s_mov_b64 exec, 1")
        e("global_store_dwordx4 v[224:225], v[226:229], off offset:16")
        e("s_mov_b64 exec, s[98:99]")
        e("L_nostamp1_%=:")
        L.extend(gen_sk_tail())

    def cline(ln):
        # the MFMA mnemonic is a macro of common.h (bf16 in libfvqa_hip.so, fp16 in libfvqa_hip_f16.so): adjacent string literals
        if ln.startswith("v_mfma_f32_16x16x32_bf16 "):
            return f'      FVQA_MFMA_H16_ASM "{ln[len("v_mfma_f32_16x16x32_bf16"):]}\\n"'
        return f'      "{ln}\\n"'
    body = "\n".join(cline(ln) for ln in L)
    accs = ", ".join(f'"+{{a[{16 * j}:{16 * j + 15}]}}"(acc[{j}])' for j in range(NBT))
    clob = ['"memory"', '"scc"', '"vcc"']
    clob += [f'"v{n}"' for n in range(64, 248)]
    clob += [f'"s{n}"' for n in range(42, 100 if sk else 96)]
    clob_s = ", ".join(clob)
    if sk:
        xp_enum = ", ".join(f"XP_{k} = {v}" for k, v in XP.items())
        return f'''
// The split-K form of the NBT = 16 loop (gemm_sk.hip gemm4w_sk_k): the loop, then — still inside the SAME statement, so that
// the compiler never sees the accumulators between them — the exchange of the tile's pieces (generator: "split-K exchange").
// xp: the exchange parameters, parameter k in lane k (XP_* below); fa: lane q < np - 1 = address of the epoch flag of partner q.
struct Ring4AsmSK {{
  enum {{ {xp_enum} }};                 // lanes of xp
  static __device__ __forceinline__ void run(f32x16 (&acc)[{NBT}], unsigned long long rsA01, unsigned long long rsB01,
                                             unsigned long long rsAB2, unsigned long long str8, unsigned long long wl,
                                             unsigned kb0, unsigned nw, unsigned voffA, unsigned voffB, unsigned rdA,
                                             unsigned rdB, unsigned xp, unsigned long long fa) {{
    asm volatile(
{body}
      : {accs}
      : [rsA01] "s"(rsA01), [rsB01] "s"(rsB01), [rsAB2] "s"(rsAB2), [str8] "s"(str8), [wl] "s"(wl), [kb0] "s"(kb0), [nw] "s"(nw),
        [voffA] "v"(voffA), [voffB] "v"(voffB), [rdA] "v"(rdA), [rdB] "v"(rdB), [xp] "v"(xp), [fa] "v"(fa)
      : {clob_s});
  }}
}};
'''
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


# ---- split-K exchange of the 4-wave kernel in the accumulator registers (NBT = 16; gemm_sk.hip: gemm4w_sk_k) ----------------
# A tile cut NP ways along K: piece c has reduced... register row blocks [0, OWN) (OWN = 4 / NP) of every wave — the row
# permutation of the loop (rowxor) puts the rows a piece owns there — and hands the blocks [OWN, 4) to its partners through its
# slab. Slab image of a workgroup (24-bit partials, gemm_sk.hip pack24): wave w at w * 64 KiB, row block i at i * 12 KiB, the
# 16 quads of a row block as 12 words of 1 KiB (64 lanes x 16 bytes): word 3*jg + k of column-block group jg.
# Both passes are generated assembly for the same reason as the loop: C++ around the accumulator vectors made hipcc spill to
# scratch (360-520 bytes per lane), and a kernel with scratch pays at every launch boundary of the step.
XP = dict(NP=0, C=1, SLAB=2, PSLAB=4, POFF=10, FLAG=13, EPOCH=15, ERR=17, WOFF=19, SLABBYTES=20, STAMP=21)   # lanes of the xp operand
V_XOFF = "v224"                                                  # per-lane slab offset: w * 64 KiB + lane * 16


def gen_publish(NP):
    """row blocks [OWN, 4) of the accumulators -> this workgroup's slab (descriptor s[44:47]), 24-bit packed"""
    OWN = 4 // NP
    L = []
    e = L.append
    e(f"; ---- publish row blocks [{OWN}, 4) of a split-K piece ({NP} pieces)")
    cnt = 0
    for i in range(OWN, 4):
        for jg in range(4):
            O = 96 + 16 * (cnt & 1)
            cnt += 1
            for jj in range(4):
                for el in range(4):
                    e(f"v_accvgpr_read_b32 v{64 + 4 * jj + el}, a{16 * (4 * jg + jj) + 4 * i + el}")
            for idx in range(16):                       # round to 24 bits unless the exponent is all ones (pack24 / round24)
                u = f"v{64 + idx}"
                e(f"v_and_b32 v80, s48, {u}")
                e(f"v_add_u32 v81, 0x80, {u}")
                e("v_cmp_ne_u32 vcc, s48, v80")
                e(f"v_cndmask_b32 {u}, {u}, v81, vcc")
            for jj in range(4):
                u0, u1, u2, u3 = (f"v{64 + 4 * jj + k}" for k in range(4))
                e(f"v_perm_b32 v{O + 3 * jj}, {u0}, {u3}, s49")
                e(f"v_perm_b32 v{O + 3 * jj + 1}, {u1}, {u3}, s50")
                e(f"v_perm_b32 v{O + 3 * jj + 2}, {u2}, {u3}, s51")
            e(f"s_mov_b32 s52, {(i * 12 + jg * 3) * 1024}")
            for k in range(3):
                off = f" offset:{k * 1024}" if k else ""
                e(f"buffer_store_dwordx4 v[{O + 4 * k}:{O + 4 * k + 3}], {V_XOFF}, s[44:47], s52 offen{off} sc1")
    return L


def gen_reduce(NP, C):
    """piece C of NP: partner q's descriptor in s[44+4q : 47+4q], its block-group offset in s[72+q]"""
    OWN = 4 // NP
    L = []
    e = L.append
    e(f"; ---- piece {C} of {NP}: fetch the partners' partials of row blocks [0, {OWN}) and add them in piece order")
    reg = 64
    where = {}
    for i in range(OWN):
        for jg in range(4):
            for q in range(NP - 1):
                e(f"s_add_u32 s56, s{72 + q}, {(i * 12 + jg * 3) * 1024}")
                where[(i, jg, q)] = reg
                for k in range(3):
                    off = f" offset:{k * 1024}" if k else ""
                    e(f"buffer_load_dwordx4 v[{reg}:{reg + 3}], {V_XOFF}, s[{44 + 4 * q}:{47 + 4 * q}], s56 offen{off} sc1")
                    reg += 4
    assert reg <= 208
    e("s_mov_b32 s57, 0xFFFFFF00")
    e("s_mov_b32 s58, 0x04000C0C")
    e("s_mov_b32 s59, 0x0706000C")
    e("s_waitcnt vmcnt(0)")
    L.extend(gen_stamp(4, f"r{NP}{C}"))                 # the partners' partials have arrived
    order = [("own", None) if pce == C else ("q", pce if pce < C else pce - 1) for pce in range(NP)]

    def unpack(base, jj, dst):
        """quad jj of a 12-register packed group at v[base..] -> 4 floats in v[dst..dst+3] (unpack24)"""
        d0, d1, d2 = (f"v{base + 3 * jj + k}" for k in range(3))
        e(f"v_perm_b32 v{dst + 3}, {d0}, {d1}, s58")
        e(f"v_perm_b32 v{dst + 3}, v{dst + 3}, {d2}, s59")
        e(f"v_and_b32 v{dst}, s57, {d0}")
        e(f"v_and_b32 v{dst + 1}, s57, {d1}")
        e(f"v_and_b32 v{dst + 2}, s57, {d2}")

    for i in range(OWN):
        for jg in range(4):
            for jj in range(4):
                a0 = 16 * (4 * jg + jj) + 4 * i
                for el in range(4):
                    e(f"v_accvgpr_read_b32 v{208 + el}, a{a0 + el}")
                first = True
                for kind, q in order:
                    if kind == "own":
                        src = 208
                    else:
                        src = 212
                        unpack(where[(i, jg, q)], jj, 212)
                    if first:
                        for el in range(4):
                            e(f"v_mov_b32 v{216 + el}, v{src + el}")
                        first = False
                    else:
                        for el in range(4):
                            e(f"v_add_f32 v{216 + el}, v{216 + el}, v{src + el}")
                for el in range(4):
                    e(f"v_accvgpr_write_b32 a{a0 + el}, v{216 + el}")
    return L


def gen_stamp(i, tag=""):
    """diagnostic (tools/sk_clock.py): 100 MHz time into word 8 + i of the workgroup's stamp record, by lane 0 of wave 0, when the
    record address s[96:97] is set (the shipping build passes 0: two scalar instructions and a branch). Clobbers SCC."""
    L = []
    e = L.append
    lab = f"L_ns{i}{tag}_%="
    e("s_cmp_eq_u64 s[96:97], 0")
    e(f"s_cbranch_scc1 {lab}")
    e("s_cmp_lg_u32 s62, 0")
    e(f"s_cbranch_scc1 {lab}")
    e("s_memrealtime s[98:99]")
    e("s_waitcnt lgkmcnt(0)")
    e("v_mov_b32 v236, s96")
    e("v_mov_b32 v237, s97")
    e("v_mov_b32 v238, s98")
    e("v_mov_b32 v239, s99")
    e("s_mov_b64 s[98:99], exec")
    e("s_mov_b64 exec, 1")
    e(f"global_store_dwordx2 v[236:237], v[238:239], off offset:{64 + 8 * i}")
    e("s_mov_b64 exec, s[98:99]")
    e(f"{lab}:")
    return L


def gen_sk_tail():
    """after the loop, inside the same statement: publish -> barrier -> flag / poll (wave 0) -> barrier -> fetch and add.
    np == 1 (a whole tile): one barrier (every wave is done reading the ring), nothing else."""
    L = []
    e = L.append

    def lane(dst, k):
        e(f"v_readlane_b32 {dst}, %[xp], {k}")

    e("; ---- split-K exchange")
    lane("s60", XP["NP"])
    lane("s61", XP["C"])
    for k in range(2):
        lane(f"s{44 + k}", XP["SLAB"] + k)
    lane("s46", XP["SLABBYTES"])
    lane("s62", XP["WOFF"])
    for q in range(3):
        lane(f"s{72 + q}", XP["POFF"] + q)
    for k in range(2):
        lane(f"s{66 + k}", XP["EPOCH"] + k)
        lane(f"s{68 + k}", XP["FLAG"] + k)
    e("s_mov_b32 s47, 0x00020000")
    e("v_mbcnt_lo_u32_b32 v224, -1, 0")
    e("v_mbcnt_hi_u32_b32 v224, -1, v224")
    e("v_lshlrev_b32 v224, 4, v224")
    e("s_nop 4")                                        # (v_readlane results -> scalar / memory instructions)
    e(f"v_add_u32 {V_XOFF}, s62, v224")
    e("s_cmp_lt_u32 s60, 2")
    e("s_cbranch_scc0 L_xchg_%=")
    e("s_barrier")
    e("s_branch L_xdone_%=")
    e("L_xchg_%=:")
    e("s_mov_b32 s48, 0x7F800000")
    e("s_mov_b32 s49, 0x07060503")
    e("s_mov_b32 s50, 0x07060502")
    e("s_mov_b32 s51, 0x07060501")
    e("s_cmp_eq_u32 s60, 2")
    e("s_cbranch_scc1 L_pub2_%=")
    L.extend(gen_publish(4))
    e("s_branch L_pubdone_%=")
    e("L_pub2_%=:")
    L.extend(gen_publish(2))
    e("L_pubdone_%=:")
    L.extend(gen_stamp(0))                              # publish issued
    e("s_waitcnt vmcnt(0)")                             # EVERY storing wave drains its write-through stores
    L.extend(gen_stamp(1))                              # ... drained
    e("s_barrier")
    L.extend(gen_stamp(2))
    # wave 0: raise this workgroup's flag (lane 0), then lanes 0 .. np-2 poll one partner's flag each (bounded)
    e("s_cmp_lg_u32 s62, 0")
    e("s_cbranch_scc1 L_flagdone_%=")
    e("s_mov_b64 s[64:65], exec")
    e("v_mov_b32 v226, s66")
    e("v_mov_b32 v227, s67")
    e("v_mov_b32 v228, s68")
    e("v_mov_b32 v229, s69")
    e("s_mov_b64 exec, 1")
    e("global_store_dwordx2 v[228:229], v[226:227], off sc1")
    e("s_sub_u32 s70, s60, 1")
    e("s_bfm_b64 exec, s70, 0")
    e("s_mov_b32 s71, 0")
    e("L_poll_%=:")
    e("global_load_dwordx2 v[230:231], %[fa], off sc1")
    e("s_waitcnt vmcnt(0)")
    e("v_cmp_ne_u64 vcc, v[226:227], v[230:231]")
    e("s_cmp_eq_u64 vcc, 0")
    e("s_cbranch_scc1 L_polled_%=")
    e("s_sleep 8")
    e("s_add_u32 s71, s71, 1")
    e("s_cmp_lt_u32 s71, 0x400000")
    e("s_cbranch_scc1 L_poll_%=")
    lane("s68", XP["ERR"])                              # a lost partner: raise the error word and go on
    lane("s69", XP["ERR"] + 1)
    e("s_mov_b64 exec, 1")
    e("s_nop 4")
    e("v_mov_b32 v228, s68")
    e("v_mov_b32 v229, s69")
    e("v_mov_b32 v232, 1")
    e("v_mov_b32 v233, 0")
    e("global_atomic_or_x2 v[228:229], v[232:233], off")
    e("s_waitcnt vmcnt(0)")
    e("L_polled_%=:")
    e("s_mov_b64 exec, s[64:65]")
    e("L_flagdone_%=:")
    e("s_barrier")
    L.extend(gen_stamp(3))                              # every partner's flag seen
    # partner descriptors: base of partner q in xp lanes PSLAB + 2q, +1
    for q in range(3):
        for k in range(2):
            lane(f"s{44 + 4 * q + k}", XP["PSLAB"] + 2 * q + k)
        if q:
            e(f"s_mov_b32 s{46 + 4 * q}, s46")
            e(f"s_mov_b32 s{47 + 4 * q}, s47")
    e("s_nop 4")
    e("s_cmp_eq_u32 s60, 2")
    e("s_cbranch_scc1 L_red2_%=")
    for c in range(4):
        if c < 3:
            e(f"s_cmp_eq_u32 s61, {c}")
            e(f"s_cbranch_scc0 L_red4n{c}_%=")
        L.extend(gen_reduce(4, c))
        e("s_branch L_reddone_%=")
        if c < 3:
            e(f"L_red4n{c}_%=:")
    e("L_red2_%=:")
    e("s_cmp_eq_u32 s61, 0")
    e("s_cbranch_scc0 L_red2n_%=")
    L.extend(gen_reduce(2, 0))
    e("s_branch L_reddone_%=")
    e("L_red2n_%=:")
    L.extend(gen_reduce(2, 1))
    e("L_reddone_%=:")
    L.extend(gen_stamp(5))                              # partials added
    e("s_nop 4")
    e("L_xdone_%=:")
    return L


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
    widths = [int(x) for x in sys.argv[1:]] or [16, 14, 13, 12, 11]       # the widths gemm4w.hip dispatches
    with open(out, "w") as f:
        f.write(HEADER)
        for n in widths:
            f.write(gen(n))
        if 16 in widths:
            f.write(gen(16, sk=True))
        f.write("\n}  // namespace fvqa_ring4\n")
    print(out)
