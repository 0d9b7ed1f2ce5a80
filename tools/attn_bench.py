#!/usr/bin/env python3
"""Attention forward / backward kernel times (bf16 build) at the benchmark shapes: C2 (8 x 128), C3 (24 x 128), C4 (3 x 650);
HIP events, median of 20. Default: the step's form — q, k arrive rotated by the QKV projection's epilogue, the forward runs
without tables and the backward un-rotates dq / dk at its store (attn_bwd(..., prerotated=True)); AB_RAW=1: the round-2 form
(raw q, k rotated inside every kernel). FVQA_ATTN_FWD32=0/1 forces the 16- / 32-query forward."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "flipped-vqa_amd"))
import torch  # noqa: E402
from fvqa import ops  # noqa: E402

dev = "cuda"
H, Dh, A, F = 32, 128, 10, 10
D = H * Dh
ONLY = os.environ.get("AB_ONLY")
for name, N, S in [("C2", 8, 128), ("C3", 24, 128), ("C4", 3, 650), ("S256", 12, 256), ("S384", 6, 384)]:
    if ONLY and name != ONLY:
        continue
    qkv = torch.randn(N * S + A, 3 * D, device=dev).bfloat16()
    o = torch.empty(N * S, D, dtype=torch.bfloat16, device=dev)
    la = torch.empty(N * H * S, dtype=torch.float32, device=dev)
    lt = torch.empty_like(la)
    g1, g2 = torch.randn(H, device=dev), torch.randn(H, device=dev) - 3
    vs = torch.full((N,), 19, dtype=torch.int32, device=dev)
    pos = torch.arange(S, device=dev, dtype=torch.float32)
    inv = 1.0 / (10000 ** (torch.arange(0, Dh, 2, device=dev).float() / Dh))
    ang = pos[:, None] * inv[None, :]
    rope = (ang.cos().contiguous(), ang.sin().contiguous())
    d_o = torch.randn(N * S, D, device=dev).bfloat16()
    dqkv = torch.empty_like(qkv)
    dg1, dg2 = torch.zeros(H, device=dev), torch.zeros(H, device=dev)
    ws = torch.zeros(ops.attn_bwd_workspace(N, S, H, Dh, A), dtype=torch.uint8, device=dev)

    def timed(fn):
        ts = []
        for r in range(24):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            fn()
            e1.record()
            torch.cuda.synchronize()
            if r >= 4:
                ts.append(e0.elapsed_time(e1) * 1e3)
        return sorted(ts)[len(ts) // 2]

    if os.environ.get("AB_RAW") == "1":
        tf = timed(lambda: ops.attn_fwd(qkv, o, la, lt, g1, g2, vs, N, S, H, Dh, A, F, rope=rope))
        tb = timed(lambda: ops.attn_bwd(d_o, qkv, o, la, lt, g1, g2, vs, dqkv, dg1, dg2, ws, N, S, H, Dh, A, F, rope=rope))
    else:
        tf = timed(lambda: ops.attn_fwd(qkv, o, la, lt, g1, g2, vs, N, S, H, Dh, A, F))
        tb = timed(lambda: ops.attn_bwd(d_o, qkv, o, la, lt, g1, g2, vs, dqkv, dg1, dg2, ws, N, S, H, Dh, A, F, rope=rope,
                                        prerotated=True))
    print(f"{name:5s} n_seq={N:3d} S={S:4d}: forward {tf:7.1f} us   backward {tb:7.1f} us", flush=True)
