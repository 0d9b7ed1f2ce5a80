#!/usr/bin/env python3
"""One projection shape through the kernel variants of fvqa_gemm_nt (0 = what the step uses, 2 = 128x128 whole-K tiles):
us per launch, median over interleaved rounds.   python tools/gemm_variants.py [M N K ...triples]"""
import math
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "flipped-vqa_amd"))
import torch  # noqa: E402
from fvqa import ops  # noqa: E402

shapes = [(1024, 4096, 4096), (1024, 4096, 11008), (1024, 4096, 12288)]
if len(sys.argv) > 3:
    v = [int(x) for x in sys.argv[1:]]
    shapes = [tuple(v[i:i + 3]) for i in range(0, len(v), 3)]
variants = [int(x) for x in os.environ.get("GV_VARIANTS", "0,2").split(",")]
for (M, N, K) in shapes:
    a = (torch.rand(M, K, device="cuda") * 2 - 1).bfloat16()
    b = ((torch.rand(N, K, device="cuda") * 2 - 1) / math.sqrt(K)).bfloat16()
    r = (torch.rand(M, N, device="cuda") * 2 - 1).bfloat16()
    out = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
    res = {}
    for resid in (False, True):
        samples = {v: [] for v in variants}
        for rnd in range(7):
            for v in variants:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(10):
                    ops.gemm_nt(a, b, out, residual=r if resid else None, variant=v)
                e1.record()
                e1.synchronize()
                if rnd:
                    samples[v].append(e0.elapsed_time(e1) * 100.0)
        res[resid] = {v: sorted(s)[len(s) // 2] for v, s in samples.items()}
    fl = 2.0 * M * N * K
    print(f"{M} x {N} x {K}: " + "  ".join(f"variant {v}: {res[False][v]:6.1f} us ({fl / res[False][v] / 1e6:5.0f} TF/s), +residual {res[True][v]:6.1f}"
                                             for v in variants), flush=True)
