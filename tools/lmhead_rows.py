#!/usr/bin/env python3
"""LM head on the scored rows only: time the candidate kernels for the compact shapes (M rows of the ~1000 a batch holds).

  forward   logits (M, V) fp32 = x (M, D) . W_out (V, D)^T
  backward  dx (M, D)          = dlogits (M, V) . W_out^T (D, V)^T

for M in --rows, through fvqa_gemm_nt: variant 0 (the dispatcher's pick), 13 (persistent / one-wave-per-SIMD family; tile width forced
with --nbt), chunks of 16 rows through the decode-shape kernel (12). Median of interleaved rounds, events around 20 launches."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "flipped-vqa_amd"))
import torch  # noqa: E402
from fvqa import ops  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--rows", default="34,66,130,258,402,1024")
ap.add_argument("--dim", type=int, default=4096)
ap.add_argument("--vocab", type=int, default=32000)
ap.add_argument("--rounds", type=int, default=5)
a = ap.parse_args()
dev = "cuda"
D, V = a.dim, a.vocab
torch.manual_seed(0)
W = (torch.randn(V, D, device=dev) * 0.02).bfloat16()
Wt = W.t().contiguous()


def timed(fn, n=20):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


for M in [int(x) for x in a.rows.split(",")]:
    x = torch.randn(M, D, device=dev).bfloat16()
    dl = (torch.randn(M, V, device=dev) * 1e-3).bfloat16()
    lg = torch.empty(M, V, dtype=torch.float32, device=dev)
    dx = torch.empty(M, D, dtype=torch.bfloat16, device=dev)
    cand = {}
    cand["fwd v0"] = lambda: ops.gemm_nt(x, W, lg)
    for nbt in (16, 12, 11):
        cand[f"fwd v13 nbt{nbt}"] = (lambda nbt=nbt: ops.gemm_nt(x, W, lg, variant=13, nbt=nbt))
    cand["fwd v13 auto"] = lambda: ops.gemm_nt(x, W, lg, variant=13)
    if M <= 130:
        def chunks_f():
            for r in range(0, M, 16):
                ops.gemm_nt(x[r:r + 16], W, lg[r:r + 16], variant=12)
        cand["fwd 16-row chunks"] = chunks_f
    cand["bwd v0"] = lambda: ops.gemm_nt(dl, Wt, dx)
    cand["bwd v13 auto"] = lambda: ops.gemm_nt(dl, Wt, dx, variant=13)
    if M <= 130:
        def chunks_b():
            for r in range(0, M, 16):
                ops.gemm_nt(dl[r:r + 16], Wt, dx[r:r + 16], variant=12)
        cand["bwd 16-row chunks"] = chunks_b
    res = {k: [] for k in cand}
    for _ in range(a.rounds):
        for k, fn in cand.items():
            try:
                res[k].append(timed(fn))
            except Exception as e:          # a shape a variant refuses
                res[k].append(float("nan"))
                err = str(e)[:60]
    ref = (x.float() @ W.float().t())
    ops.gemm_nt(x, W, lg, variant=13)
    e13 = float((lg - ref).abs().max() / ref.abs().max())
    ops.gemm_nt(x, W, lg)
    e0 = float((lg - ref).abs().max() / ref.abs().max())
    print(f"M={M:5d}  (rel err v13 {e13:.1e} v0 {e0:.1e})  " + "  ".join(f"{k}: {sorted(v)[len(v) // 2]:.1f}" for k, v in res.items()), flush=True)
