#!/usr/bin/env python3
"""Decode-shape (M = 8) projection micro-benchmark: weight bytes / time of the skinny kernel per 7B shape."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "flipped-vqa_amd"))
import torch  # noqa: E402
from fvqa import ops  # noqa: E402

M = int(os.environ.get("SB_M", "8"))
tot_t = tot_b = 0.0
for name, N, K in [("qkv", 12288, 4096), ("wo", 4096, 4096), ("w13", 22016, 4096), ("w2", 4096, 11008),
                   ("head", 32000, 4096)]:
    a = torch.randn(M, K, device="cuda").bfloat16()
    w = (torch.randn(N, K, device="cuda") / K ** 0.5).bfloat16()
    out = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
    ts = []
    for r in range(12):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        ops.gemm_nt(a, w, out)
        e1.record()
        torch.cuda.synchronize()
        if r >= 2:
            ts.append(e0.elapsed_time(e1) * 1e3)
    t = sorted(ts)[len(ts) // 2]
    if name != "head":
        tot_t += t
        tot_b += N * K * 2
    print(f"{name:5s} N={N:6d} K={K:6d}: {t:7.1f} us  {N * K * 2 / t / 1e6:5.2f} TB/s", flush=True)
print(f"layer: {tot_t:.1f} us, {tot_b / tot_t / 1e6:.2f} TB/s of weight bytes")
