#!/usr/bin/env python3
"""How does the old split-K path scale with the number of splits (active CUs) on the N = 4096 outputs, and what do
zero operands (no data-dependent switching power) change? Tuning probe."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "flipped-vqa_amd"))
import torch  # noqa: E402
from fvqa import ops  # noqa: E402

dev = "cuda"


def timed(fn, reps=20, rounds=5):
    ts = []
    for r in range(rounds + 1):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        if r:
            ts.append(e0.elapsed_time(e1) * 1e3 / reps)
    ts.sort()
    return ts[len(ts) // 2]


for name, M, N, K in [("wo", 1024, 4096, 4096), ("w2", 1024, 4096, 11008), ("w13t", 1024, 4096, 22016),
                      ("qkv", 1024, 12288, 4096), ("head", 1024, 32000, 4096)]:
    for fill in ("rand", "zero"):
        if fill == "rand":
            a = (torch.rand(M, K, device=dev) * 2 - 1).bfloat16()
            b = ((torch.rand(N, K, device=dev) * 2 - 1) / K ** 0.5).bfloat16()
        else:
            a = torch.zeros(M, K, device=dev, dtype=torch.bfloat16)
            b = torch.zeros(N, K, device=dev, dtype=torch.bfloat16)
        o = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
        cells = []
        for v in (17, 18, 19, 20, 13):
            if N > 4096 and v in (18, 19, 20):
                continue
            t = timed(lambda: ops.gemm_nt(a, b, o, variant=v))
            cells.append(f"v{v}: {t:6.1f} us {2.0 * M * N * K / t / 1e6:5.0f} TF")
        print(f"{name:5s} {fill}  " + " | ".join(cells), flush=True)
