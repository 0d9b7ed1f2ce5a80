#!/usr/bin/env python3
"""Phase timeline of the fused attention backward (S <= 128) from its in-kernel stamps (tuning build
-DFVQA_ATTN_STAMPS; run with FVQA_LIB=<that build>): per workgroup, 100 MHz timestamps at kernel start, operands staged,
after pass A (dQ), after pass B (dK, dV), after pass C (adapter keys), partials published, arrival known, reducer done."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "flipped-vqa_amd"))
import numpy as np  # noqa: E402
import torch  # noqa: E402
from fvqa import ops, _lib  # noqa: E402

dev = "cuda"
A, F, Dh = 10, 10, 128
N, S, H = int(os.environ.get("AB_N", "8")), 128, 32
D = H * Dh
torch.manual_seed(0)
qkv = (torch.randn(N * S + A, 3 * D, device=dev) * 0.5).bfloat16()
d_o = torch.randn(N * S, D, device=dev).bfloat16()
g1 = torch.randn(H, device=dev) * 0.5
g2 = torch.full((H,), -3.5, device=dev)
vs = torch.full((N,), 19, dtype=torch.int32, device=dev)
ang = torch.outer(torch.arange(2 * S, device=dev, dtype=torch.float32),
                  1.0 / (10000.0 ** (torch.arange(0, Dh, 2, device=dev).float() / Dh)))
rope = (ang.cos().contiguous(), ang.sin().contiguous())
# the step's form by default: q, k arrive rotated (RoPE in the QKV projection's epilogue), the forward runs without tables and
# the backward un-rotates dq / dk at its store; AB_RAW=1: the round-2 form (raw q, k rotated inside every kernel)
RAW = os.environ.get("AB_RAW") == "1"
FWD_KW = dict(rope=rope) if RAW else {}
BWD_KW = dict(rope=rope) if RAW else dict(rope=rope, prerotated=True)
o = torch.empty(N * S, D, dtype=torch.bfloat16, device=dev)
la = torch.empty(N * H * S, device=dev)
lt = torch.empty_like(la)
dqkv = torch.empty_like(qkv)
dg1, dg2 = torch.zeros(H, device=dev), torch.zeros(H, device=dev)
ws = torch.zeros(ops.attn_bwd_workspace(N, S, H, Dh, A), dtype=torch.uint8, device=dev)
lib = _lib.load()
raw = ctypes.CDLL(_lib.lib_path())
if os.environ.get("AB_FWD"):
    # forward kernel at S = AB_S (default 650), n_seq = AB_N (default 3): per workgroup, time of wave 0 / wave 7 per phase
    S = int(os.environ.get("AB_S", "650"))
    N = int(os.environ.get("AB_N", "3"))
    qkv = (torch.randn(N * S + A, 3 * D, device=dev) * 0.5).bfloat16()
    vs = torch.full((N,), 19, dtype=torch.int32, device=dev)
    ang = torch.outer(torch.arange(2 * S, device=dev, dtype=torch.float32),
                      1.0 / (10000.0 ** (torch.arange(0, Dh, 2, device=dev).float() / Dh)))
    rope = (ang.cos().contiguous(), ang.sin().contiguous())
    o = torch.empty(N * S, D, dtype=torch.bfloat16, device=dev)
    la = torch.empty(N * H * S, device=dev)
    lt = torch.empty_like(la)
    raw.fvqa_attn_stamps_read.argtypes = [ctypes.c_void_p, ctypes.c_int]
    buf = np.zeros(1024 * 16, dtype=np.uint64)
    for _ in range(5):
        ops.attn_fwd(qkv, o, la, lt, g1, g2, vs, N, S, H, Dh, A, F, **FWD_KW)
    torch.cuda.synchronize()
    raw.fvqa_attn_stamps_read(buf.ctypes.data, 1)
    ops.attn_fwd(qkv, o, la, lt, g1, g2, vs, N, S, H, Dh, A, F, **FWD_KW)
    torch.cuda.synchronize()
    raw.fvqa_attn_stamps_read(buf.ctypes.data, 0)
    nqb = (S + 127) // 128
    st = buf.reshape(1024, 16)[: min(1024, N * H * nqb)].astype(np.float64)
    st = st[st[:, 0] > 0]
    t0 = st[:, 0].min()
    print(f"attn_fwd n_seq={N} S={S} H={H}: {len(st)} workgroups stamped; kernel span {(st[:, 5].max() - t0) / 100:.1f} us; "
          f"last start {(st[:, 0].max() - t0) / 100:.1f} us")
    for qb in range(nqb):
        for off, wn in ((0, "wave 0"), (8, "wave 7")):
            r = st[st[:, 7] == qb]
            if not len(r):
                continue
            f = lambda c: np.median(r[:, off + c]) / 100.0
            print(f"  query block {qb} ({qb + 1} tiles) {wn}: life {np.median(r[:, off + 5] - r[:, off + 0]) / 100:6.1f} us = "
                  f"commit {f(1):5.1f} + barrier {f(2):5.1f} + key groups {f(3):5.1f} + barrier {f(4):5.1f} + rest")
    sys.exit(0)
raw.fvqa_attn_stamps_read.argtypes = [ctypes.c_void_p, ctypes.c_int]
buf = np.zeros(1024 * 16, dtype=np.uint64)
ops.attn_fwd(qkv, o, la, lt, g1, g2, vs, N, S, H, Dh, A, F, **FWD_KW)
for _ in range(5):
    ops.attn_bwd(d_o, qkv, o, la, lt, g1, g2, vs, dqkv, dg1, dg2, ws, N, S, H, Dh, A, F, **BWD_KW)
torch.cuda.synchronize()
raw.fvqa_attn_stamps_read(buf.ctypes.data, 1)
ops.attn_bwd(d_o, qkv, o, la, lt, g1, g2, vs, dqkv, dg1, dg2, ws, N, S, H, Dh, A, F, **BWD_KW)
torch.cuda.synchronize()
raw.fvqa_attn_stamps_read(buf.ctypes.data, 0)
st = buf.reshape(1024, 16)[: N * H].astype(np.float64)
t0 = st[:, 0].min()
names = ["start", "staged", "pass A", "pass B", "pass C", "published", "arrival", "reduced"]
print(f"attn_bwd_fused n_seq={N} S={S} H={H}: us since the first workgroup started (median / min / max over workgroups)")
for k, nm in enumerate(names):
    col = st[:, k]
    col = col[col > 0]
    if col.size:
        us = (col - t0) / 100.0
        print(f"  {nm:10s} n={col.size:4d}  med {np.median(us):6.1f}  min {us.min():6.1f}  max {us.max():6.1f}")
