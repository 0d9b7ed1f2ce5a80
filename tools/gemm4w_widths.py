#!/usr/bin/env python3
"""Does the tile-width cost model of the whole-tile projection kernel pick the fastest width?  (round-4 verdict #5)

For every projection of BASELINE configs[1..4] (C2 .. C5) and of the reference's S = 256 / S = 384 recipes (README.md:70-88)
that the 4-wave whole-tile kernel takes (csrc/gemm4w.hip: q|k|v + RoPE, W1|W3 + SwiGLU with the adapter K/V rider, dH W2^T +
SwiGLU' with the adapter-gradient rider, the LM head, and — at 1536 rows and more — the N = dim outputs with and without the
residual), every legal width (16 * nbt columns, nbt in 11, 12, 13, 14, 16; even only for the SwiGLU-forward epilogue) is timed
through ops.gemm4w_width, interleaved round-robin on one device (rule 24: deltas from interleaved rounds in one process;
median over rounds of the mean of `reps` back-to-back launches between two events), next to the width `fvqa_gemm4w_choose`
picks. A pick more than `--tol` slower than the best width is flagged.

    python tools/gemm4w_widths.py [--rounds 5 --reps 8 --tol 0.03 --configs c2,c3,c4,c5,s256,s384] > profiles/r05_gemm4w_widths.log

tests/test_kernels_gpu.py::test_gemm4w_chooser_picks_a_width_within_tolerance runs `survey()` on the C2 shapes.
"""
import argparse
import math
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "flipped-vqa_amd"))
import torch  # noqa: E402
from fvqa import ops  # noqa: E402

WIDTHS = (16, 14, 13, 12, 11)
EPI_NONE, EPI_RESIDUAL, EPI_SWIGLU_FWD_ST, EPI_SWIGLU_BWD_ST, EPI_ROPE = 0, 1, 5, 6, 7

# name -> (rows = sequences x seq_len over the batched streams, seq_len, dim, hidden, heads)
CONFIGS = {
    "c2": (8 * 128, 128, 4096, 11008, 32),           # 7B, B = 8, VQA only
    "c3": (24 * 128, 128, 4096, 11008, 32),          # 7B, B = 8, three streams
    "c4": (3 * 650, 650, 4096, 11008, 32),           # 7B, S = 650, B = 1, three streams
    "c5": (12 * 128, 128, 5120, 13824, 40),          # 13B, B = 4, three streams
    "s256": (12 * 256, 256, 4096, 11008, 32),        # VLEP recipe: S = 256, B = 4, three streams
    "s384": (6 * 384, 384, 4096, 11008, 32),         # DramaQA recipe: S = 384, B = 2, three streams
}
V = 32000
A = 10


def projections(R, S, D, Hf, H):
    """(label, kind, M, N, K) of the step's projections; kind selects the call form below"""
    return [("qkv+rope", "rope", R, 3 * D, D), ("w13+swiglu", "swf", R, 2 * Hf, D), ("w2t+swiglu'", "swb", R, Hf, D),
            ("lm_head", "f32", R, V, D), ("wo+res", "res", R, D, D), ("w2+res", "res", R, D, Hf),
            ("wo_t", "none", R, D, D), ("qkv_t", "none", R, D, 3 * D), ("w13_t", "none", R, D, 2 * Hf), ("lm_head_t", "none", R, D, V)]


def rnd(*shape, scale=1.0):
    return (torch.rand(*shape, device="cuda") * 2 - 1).mul_(scale).to(torch.bfloat16)


class Case:
    def __init__(self, label, kind, M, N, K, S, D, H):
        self.label, self.kind, self.M, self.N, self.K = label, kind, M, N, K
        self.a = rnd(M, K)
        self.b = rnd(N, K, scale=1 / math.sqrt(K))
        self.rider_nk = None
        if kind == "rope":
            Dh = D // H                               # RoPE tables (llama/model.py:45-50), rows of Dh / 2 floats per position
            ang = torch.outer(torch.arange(2 * S).float(), 1.0 / (10000.0 ** (torch.arange(0, Dh, 2).float() / Dh)))
            self.rope = (torch.cos(ang).cuda(), torch.sin(ang).cuda())
            self.S, self.Dh, self.H = S, D // H, H
            self.out = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
            self.epi = EPI_ROPE
        elif kind == "swf":
            self.out = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
            self.z = torch.empty(M, N // 2, dtype=torch.bfloat16, device="cuda")
            self.ra, self.rb = rnd(A, K), rnd(2 * K, K, scale=1 / math.sqrt(K))        # next layer's adapter K/V rows
            self.ro = torch.empty(A, 2 * K, dtype=torch.bfloat16, device="cuda")
            self.rider_nk, self.epi = (2 * K, K), EPI_SWIGLU_FWD_ST
        elif kind == "swb":
            self.st = rnd(M, 2 * N)
            self.out = torch.empty(M, 2 * N, dtype=torch.bfloat16, device="cuda")
            self.ra, self.rb = rnd(A, 2 * K), rnd(K, 2 * K, scale=1 / math.sqrt(2 * K))  # adapter-gradient rows (fp32 +=)
            self.ro = torch.zeros(A, K, dtype=torch.float32, device="cuda")
            self.rider_nk, self.epi = (K, 2 * K), EPI_SWIGLU_BWD_ST
        elif kind == "f32":
            self.out = torch.empty(M, N, dtype=torch.float32, device="cuda")
            self.epi = EPI_NONE
        else:
            self.out = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
            self.res = rnd(M, N) if kind == "res" else None
            self.epi = EPI_RESIDUAL if kind == "res" else EPI_NONE

    def pick(self):
        return ops.gemm4w_choose(self.M, self.N, self.K, out_f32=self.kind == "f32", epilogue=self.epi, rider_nk=self.rider_nk)

    def launch(self):
        if self.kind == "rope":
            ops.gemm_nt_rope(self.a, self.b, self.out, self.rope, self.S, self.Dh, self.H)
        elif self.kind == "swf":
            if os.environ.get("GW_NO_RIDER") == "1":
                ops.gemm_nt_swiglu_fwd(self.a, self.b, self.out, self.z, st=True)
            else:
                ops.gemm_nt_swiglu_fwd(self.a, self.b, self.out, self.z, st=True, rider_a=self.ra, rider_b=self.rb, rider_out=self.ro)
        elif self.kind == "swb":
            if os.environ.get("GW_NO_RIDER") == "1":        # what the launch costs without its side job (what a faster rider could win)
                ops.gemm_nt_swiglu_bwd(self.a, self.b, self.st, self.out, st=True)
            else:
                ops.gemm_nt_rider(self.a, self.b, self.out, rider_a=self.ra, rider_b=self.rb, rider_out=self.ro, accumulate=True,
                                  swiglu_ab=self.st, swiglu_st=True)
        else:
            ops.gemm_nt(self.a, self.b, self.out, residual=getattr(self, "res", None))


def whole_tile(case):
    """True when the launch goes to gemm4w_k (launch record: bit 7 set, bit 4 — split-K — clear)"""
    ops.gemm_timing_enable(True, 1)
    try:
        case.launch()
        kinds = [k for (_, _, k) in ops.gemm_timing_read()]
    finally:
        ops.gemm_timing_enable(False)
    return bool(kinds) and bool(kinds[0] & 128) and not kinds[0] & 16


def time_us(case, reps):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        case.launch()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


def survey(configs, rounds=5, reps=8, tol=0.03, out=sys.stdout):
    """-> list of dicts (config, label, M, N, K, pick, best, times {nbt: us}, pick_over_best); prints one table row each"""
    rows = []
    for cname in configs:
        R, S, D, Hf, H = CONFIGS[cname]
        for (label, kind, M, N, K) in projections(R, S, D, Hf, H):
            case = Case(label, kind, M, N, K, S, D, H)
            pick = case.pick()
            if not pick or not whole_tile(case):
                del case
                torch.cuda.empty_cache()
                continue                                # a split-K launch (the N = dim outputs at R = 1024): not this kernel's
            widths = [n for n in WIDTHS if not (kind == "swf" and n % 2)]
            samples = {n: [] for n in widths}
            for n in widths:                            # warm-up of every width
                with ops.gemm4w_width(n):
                    case.launch()
            for _ in range(rounds):
                for n in widths:                        # interleaved: every round visits every width
                    with ops.gemm4w_width(n):
                        samples[n].append(time_us(case, reps))
            times = {n: sorted(v)[len(v) // 2] for n, v in samples.items()}
            best = min(times, key=times.get)
            ratio = times[pick] / times[best]
            rows.append(dict(config=cname, label=label, M=M, N=N, K=K, pick=pick, best=best, times=times, pick_over_best=ratio))
            flag = "" if ratio <= 1 + tol else f"   <-- pick {100 * (ratio - 1):.1f} % slower than nbt {best}"
            print(f"{cname:5s} {label:12s} {M:5d} x {N:6d} x {K:6d}  pick {pick:2d} best {best:2d}  " +
                  "  ".join(f"{n}: {times[n]:7.1f}" for n in widths) + f"  us   pick/best {ratio:.3f}{flag}", file=out, flush=True)
            del case
            torch.cuda.empty_cache()
    return rows


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--reps", type=int, default=8)
    ap.add_argument("--tol", type=float, default=0.03)
    ap.add_argument("--configs", default="c2,c3,c4,c5,s256,s384")
    a = ap.parse_args()
    print(f"# tools/gemm4w_widths.py: us per launch (median over {a.rounds} interleaved rounds of the mean of {a.reps} back-to-back "
          f"launches incl. their boundaries), every legal tile width of gemm4w_k per projection; pick = fvqa_gemm4w_choose")
    rows = survey(a.configs.split(","), a.rounds, a.reps, a.tol)
    bad = [r for r in rows if r["pick_over_best"] > 1 + a.tol]
    lost = sum(r["times"][r["pick"]] - r["times"][r["best"]] for r in rows)
    print(f"# {len(rows)} whole-tile projections, {len(bad)} picks more than {100 * a.tol:.0f} % off the best width; "
          f"sum over the rows of (pick - best) = {lost:.1f} us")


if __name__ == "__main__":
    main()
