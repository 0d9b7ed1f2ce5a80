#!/usr/bin/env python3
"""What the last layer costs after its attention when only M rows matter (the rows a head reads): the six projections of the
post-attention half of a layer — WO + residual, W1|W3 + SwiGLU, W2 + residual, dH.W2^T + SwiGLU', W1|W3^T, WO^T — and the four row
kernels, at M rows against the dense row count, through the same entry points the step uses."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "flipped-vqa_amd"))
import torch  # noqa: E402
from fvqa import ops  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--rows", default="33,1024")
ap.add_argument("--dim", type=int, default=4096)
ap.add_argument("--hidden", type=int, default=11008)
a = ap.parse_args()
dev, bf = "cuda", torch.bfloat16
D, Hf = a.dim, a.hidden
torch.manual_seed(0)
r = lambda *s: (torch.randn(*s, device=dev) * 0.02).to(bf)      # noqa: E731
wo, wo_t, w13, w13_t, w2, w2_t = r(D, D), r(D, D), r(2 * Hf, D), r(D, 2 * Hf), r(D, Hf), r(Hf, D)
fn = torch.ones(D, device=dev, dtype=bf)


def timed(fn_, n=20):
    fn_()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn_()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


for M in [int(x) for x in a.rows.split(",")]:
    x, o, h, hn, z, xl = r(M, D), r(M, D), r(M, D), r(M, D), r(M, Hf), r(M, D)
    ab, dab, t = r(M, 2 * Hf), r(M, 2 * Hf), r(M, D)
    rstd = torch.ones(M, device=dev)
    cand = {
        "WO+res": lambda: ops.gemm_nt(o, wo, h, residual=x),
        "norm": lambda: ops.rmsnorm_fwd(h, fn, hn, rstd, 1e-6, rows=M),
        "W13+swiglu": lambda: ops.gemm_nt_swiglu_fwd(hn, w13, ab, z, st=True),
        "W2+res": lambda: ops.gemm_nt(z, w2, xl, residual=h),
        "W2t+swiglu'": lambda: ops.gemm_nt_swiglu_bwd(xl, w2_t, ab, dab, st=True),
        "W13t": lambda: ops.gemm_nt(dab, w13_t, t),
        "norm_bwd": lambda: ops.rmsnorm_bwd(t, h, fn, rstd, hn, resid=xl, rows=M),
        "WOt": lambda: ops.gemm_nt(hn, wo_t, o),
    }
    res = {k: [] for k in cand}
    for _ in range(5):
        for k, f in cand.items():
            res[k].append(timed(f))
    med = {k: sorted(v)[len(v) // 2] for k, v in res.items()}
    print(f"M={M:5d}  " + "  ".join(f"{k}: {v:.1f}" for k, v in med.items()) + f"   sum {sum(med.values()):.1f} us", flush=True)
