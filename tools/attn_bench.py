#!/usr/bin/env python3
"""Attention micro-benchmark on the MI355X at the step's shapes (tuning aid; not part of the product path):
forward and backward of the adapter-gated attention, HIP-event timed, bf16 MFMA build with fused RoPE."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "flipped-vqa_amd"))
import torch  # noqa: E402
from fvqa import ops  # noqa: E402

dev = "cuda"
SHAPES = [("C2 7B B=8 S=128", 8, 128, 32), ("C3 3 streams", 24, 128, 32), ("C4 S=650 B=1 x3", 3, 650, 32)]
A, F, Dh = 10, 10, 128
ROUNDS = int(os.environ.get("AB_ROUNDS", "20"))
torch.manual_seed(0)
for name, N, S, H in SHAPES:
    D = H * Dh
    qkv = (torch.randn(N * S + A, 3 * D, device=dev) * 0.5).bfloat16()
    d_o = torch.randn(N * S, D, device=dev).bfloat16()
    g1 = torch.randn(H, device=dev) * 0.5
    g2 = torch.full((H,), -3.5, device=dev)
    vs = torch.full((N,), 19, dtype=torch.int32, device=dev)
    pos = torch.arange(2 * S, device=dev, dtype=torch.float32)
    inv = 1.0 / (10000.0 ** (torch.arange(0, Dh, 2, device=dev).float() / Dh))
    ang = torch.outer(pos, inv)
    cos, sin = ang.cos().contiguous(), ang.sin().contiguous()
    o = torch.empty(N * S, D, dtype=torch.bfloat16, device=dev)
    la = torch.empty(N * H * S, device=dev)
    lt = torch.empty_like(la)
    dqkv = torch.empty_like(qkv)
    dg1, dg2 = torch.zeros(H, device=dev), torch.zeros(H, device=dev)
    ws = torch.zeros(ops.attn_bwd_workspace(N, S, H, Dh, A), dtype=torch.uint8, device=dev)
    rope = (cos, sin) if ops.attn_rope_fused(torch.bfloat16) else None
    tf, tb = [], []
    for r in range(ROUNDS + 2):
        e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        e[0].record()
        ops.attn_fwd(qkv, o, la, lt, g1, g2, vs, N, S, H, Dh, A, F, rope=rope)
        e[1].record()
        ops.attn_bwd(d_o, qkv, o, la, lt, g1, g2, vs, dqkv, dg1, dg2, ws, N, S, H, Dh, A, F, rope=rope)
        e[2].record()
        torch.cuda.synchronize()
        if r >= 2:
            tf.append(e[0].elapsed_time(e[1]) * 1e3)
            tb.append(e[1].elapsed_time(e[2]) * 1e3)
    med = lambda x: sorted(x)[len(x) // 2]
    io_f = (qkv.numel() + o.numel()) * 2
    io_b = (2 * qkv.numel() + 2 * o.numel()) * 2
    print(f"{name:18s} fwd {med(tf):7.1f} us ({io_f / med(tf) / 1e6:5.2f} TB/s of its I/O)   "
          f"bwd {med(tb):7.1f} us ({io_b / med(tb) / 1e6:5.2f} TB/s)", flush=True)
