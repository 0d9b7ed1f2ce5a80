#!/usr/bin/env python3
"""Phase timeline of the persistent GEMM from its in-kernel stamps (tuning build -DFVQA_SK_STAMPS; run with
FVQA_LIB=<that build>): per workgroup, 100 MHz timestamps at kernel start and, per segment, after the ring loop,
after the slab publish, after the peers' flags were seen, after the reduction, after the tile store."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "flipped-vqa_amd"))
import torch  # noqa: E402
from fvqa import ops, _lib  # noqa: E402

dev = "cuda"
SHAPES = [("qkv_fwd", 1024, 12288, 4096), ("wo_fwd", 1024, 4096, 4096), ("w13_fwd", 1024, 22016, 4096), ("w2_fwd", 1024, 4096, 11008), ("qkvt_bwd", 1024, 4096, 12288),
          ("w2t_bwd", 1024, 11008, 4096), ("w13t_bwd", 1024, 4096, 22016)]
lib = _lib.load()
need = int(lib.fvqa_gemm_sk_workspace())
for name, M, N, K in SHAPES:
    a = (torch.rand(M, K, device=dev) * 2 - 1).bfloat16()
    b = ((torch.rand(N, K, device=dev) * 2 - 1) / K ** 0.5).bfloat16()
    o = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    for _ in range(3):
        ops.gemm_nt(a, b, o, variant=13)
    ws = ops.gemm_workspace(a.device, need)
    ws[need - 256 * 16 * 8:need].zero_()
    ops.gemm_nt(a, b, o, variant=13)
    torch.cuda.synchronize()
    st = ws[need - 256 * 16 * 8:need].view(torch.int64).view(256, 16).cpu().double()
    used = st[:, 0] > 0
    st = st[used]
    t0 = st[:, 0].min()
    rel = (st - t0) / 100.0          # us
    rel[st == 0] = float("nan")
    print(f"\n{name} {M}x{N}x{K}: {int(used.sum())} workgroups; us since the first workgroup started "
          f"(median / min / max over workgroups)")
    lab = ["start"] + [f"s{s}:{p}" for s in range(3) for p in ("loop", "publ", "seen", "redu", "stor")]
    for i, l in enumerate(lab):
        col = rel[:, i]
        col = col[~torch.isnan(col)]
        if col.numel():
            print(f"  {l:8s} n={col.numel():3d}  med {col.median():7.1f}  min {col.min():7.1f}  max {col.max():7.1f}")
