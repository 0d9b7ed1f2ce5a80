#!/usr/bin/env python3
"""Persistent GEMM (variant 13) on the MI355X: correctness against torch (fp32 accumulate), bitwise repeatability,
the error word of the workspace, and timing next to the vendor library on the same operands (tuning aid)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "flipped-vqa_amd"))
import torch  # noqa: E402
from fvqa import ops, _lib  # noqa: E402

dev = "cuda"
M0 = int(os.environ.get("GB_M", "1024"))
SHAPES = [("qkv_fwd", M0, 12288, 4096), ("wo_fwd", M0, 4096, 4096), ("w13_fwd", M0, 22016, 4096),
          ("w2_fwd", M0, 4096, 11008), ("w2t_bwd", M0, 11008, 4096), ("w13t_bwd", M0, 4096, 22016),
          ("qkvt_bwd", M0, 4096, 12288), ("head_fwd", M0, 32000, 4096), ("headt_bwd", M0, 4096, 32000),
          ("small", 300, 768, 2112), ("ragged", 1034, 512, 1024)]
ROUNDS = int(os.environ.get("GB_ROUNDS", "7"))
torch.manual_seed(0)
lib = _lib.load()


def timed(fn, reps=20):
    """median / min over rounds of the average of `reps` back-to-back launches (as the step issues them: the launch
    overhead of a lone launch after a synchronise, ~10 us, is not part of the kernel)."""
    ts = []
    for r in range(ROUNDS + 1):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        if r:
            ts.append(e0.elapsed_time(e1) * 1e3 / reps)
    ts.sort()
    return ts[len(ts) // 2], ts[0]


print(f"{'shape':10s} {'M':>5s} {'N':>6s} {'K':>6s} | persistent: us(med/min) TF err rep | torch.matmul+add (hipBLASLt): us TF")
tot_new = tot_old = 0.0
for name, M, N, K in SHAPES:
    a = (torch.rand(M, K, device=dev) * 2 - 1).bfloat16()
    b = ((torch.rand(N, K, device=dev) * 2 - 1) / K ** 0.5).bfloat16()
    r = (torch.rand(M, N, device=dev) * 2 - 1).bfloat16()
    ref = a.float() @ b.float().T + r.float()
    o1 = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    o2 = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    ops.gemm_nt(a, b, o1, residual=r, variant=13)
    ops.gemm_nt(a, b, o2, residual=r, variant=13)
    torch.cuda.synchronize()
    err = float((o1.float() - ref).abs().max() / ref.abs().max())
    rep = bool(torch.equal(o1, o2))
    t_new, t_new_min = timed(lambda: ops.gemm_nt(a, b, o1, residual=r, variant=13))
    bt = b.T
    t_old, _ = timed(lambda: torch.addmm(r, a, bt))
    if M == M0 and name not in ("head_fwd", "headt_bwd"):
        tot_new += t_new
        tot_old += t_old
    fl = 2.0 * M * N * K
    print(f"{name:10s} {M:5d} {N:6d} {K:6d} | {t_new:7.1f} {t_new_min:7.1f} {fl / t_new / 1e6:6.0f} {err:8.1e} {rep} | "
          f"{t_old:7.1f} {fl / t_old / 1e6:6.0f}", flush=True)
ws = ops.gemm_workspace(torch.device(dev, torch.cuda.current_device()), 8)
print("error word:", int(ws[:8].view(torch.int64)[0]), " per-layer sum new/old us:", round(tot_new, 1), round(tot_old, 1))

# fp32 out + fp32 build + swiglu epilogue
for (M, N, K) in ((1024, 4096, 4096), (1024, 11008, 4096), (522, 1536, 512)):
    a = (torch.rand(M, K, device=dev) * 2 - 1)
    b = ((torch.rand(N, K, device=dev) * 2 - 1) / K ** 0.5)
    ref = a.double() @ b.double().T
    o = torch.empty(M, N, device=dev)
    ops.gemm_nt(a, b, o, variant=13)
    e32 = float((o.double() - ref).abs().max() / ref.abs().max())
    ab16, bb16 = a.bfloat16(), b.bfloat16()
    o32 = torch.empty(M, N, device=dev)
    ops.gemm_nt(ab16, bb16, o32, variant=13)
    ref16 = ab16.double() @ bb16.double().T
    e16 = float((o32.double() - ref16).abs().max() / ref16.abs().max())
    print(f"fp32 build {M}x{N}x{K}: err {e32:.1e};  bf16->fp32 out err {e16:.1e}")
