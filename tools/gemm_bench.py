#!/usr/bin/env python3
"""GEMM micro-benchmark on the MI355X: every projection shape of the 7B step (C2: M=1024+10 rows),
each kernel variant, checked against torch.matmul (fp32 accumulate) on random data and timed with
HIP events in interleaved rounds in one process (tuning aid; not part of the product path)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "flipped-vqa_amd"))
import torch  # noqa: E402
from fvqa import ops, _lib  # noqa: E402

dev = "cuda"
M0 = int(os.environ.get("GB_M", "1024"))
SHAPES = [  # name, M, N, K
    ("qkv_fwd", M0 + 10, 12288, 4096), ("wo_fwd", M0, 4096, 4096), ("w13_fwd", M0, 22016, 4096),
    ("w2_fwd", M0, 4096, 11008), ("w2t_bwd", M0, 11008, 4096), ("w13t_bwd", M0, 4096, 22016),
    ("wot_bwd", M0, 4096, 4096), ("qkvt_bwd", M0 + 10, 4096, 12288), ("head_fwd", M0, 32000, 4096),
    ("headt_bwd", M0, 4096, 32000),
]
VARIANTS = [int(v) for v in os.environ.get("GB_VARIANTS", "2,3").split(",")]
ROUNDS = int(os.environ.get("GB_ROUNDS", "5"))
lib = _lib.load()
torch.manual_seed(0)
print(f"{'shape':10s} {'M':>5s} {'N':>6s} {'K':>6s} " + " ".join(f"v{v:<3d}(us / TF / err)      " for v in VARIANTS))
tot = {v: 0.0 for v in VARIANTS}
for name, M, N, K in SHAPES:
    a = (torch.rand(M, K, device=dev) * 2 - 1).bfloat16()
    b = ((torch.rand(N, K, device=dev) * 2 - 1) / K ** 0.5).bfloat16()
    ref = a.float() @ b.float().T
    outs = {v: torch.empty(M, N, dtype=torch.bfloat16, device=dev) for v in VARIANTS}
    times = {v: [] for v in VARIANTS}
    for r in range(ROUNDS + 1):
        for v in VARIANTS:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            ops.gemm_nt(a, b, outs[v], variant=v)
            e1.record()
            torch.cuda.synchronize()
            if r:
                times[v].append(e0.elapsed_time(e1) * 1e3)
    cells = []
    for v in VARIANTS:
        t = sorted(times[v])[len(times[v]) // 2]
        err = float((outs[v].float() - ref).abs().max() / ref.abs().max())
        tot[v] += t
        sp = lib.fvqa_gemm_splits(M, N, K, 1) if v in (0, 3) else (v - 16 if v >= 16 else 1)
        cells.append(f"{t:8.1f} {2.0 * M * N * K / t / 1e6:7.0f} {err:8.1e} s{sp}")
    if os.environ.get("GB_TORCH") == "1":          # vendor library (hipBLASLt through torch) on the same operands
        ts = []
        bt = b.T
        for r in range(ROUNDS + 1):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            torch.matmul(a, bt)
            e1.record()
            torch.cuda.synchronize()
            if r:
                ts.append(e0.elapsed_time(e1) * 1e3)
        t = sorted(ts)[len(ts) // 2]
        cells.append(f"| torch.matmul {t:8.1f} us {2.0 * M * N * K / t / 1e6:7.0f} TF")
    print(f"{name:10s} {M:5d} {N:6d} {K:6d} " + "  ".join(cells), flush=True)
print("sum per layer-equivalent (us):", {v: round(t, 1) for v, t in tot.items()})

if os.environ.get("GB_STEP", "1") == "1":
    # the launches of one 7B layer exactly as the step issues them (split-K outputs left as fp32
    # partials for the fused norm kernels, SwiGLU' epilogue on w2t, tail plan on w13)
    print("\n-- one layer as the step issues it --")
    layer = [("qkv_fwd", "plain"), ("wo_fwd", "partial"), ("w13_fwd", "plain"), ("w2_fwd", "partial"),
             ("w2t_bwd", "swiglu"), ("w13t_bwd", "partial"), ("wot_bwd", "partial"), ("qkvt_bwd", "partial")]
    shp = {n: (M, N, K) for n, M, N, K in SHAPES}
    tot_t = tot_f = 0.0
    for name, kind in layer:
        M, N, K = shp[name]
        a = (torch.rand(M, K, device=dev) * 2 - 1).bfloat16()
        b = ((torch.rand(N, K, device=dev) * 2 - 1) / K ** 0.5).bfloat16()
        out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
        if kind == "swiglu":
            ab = (torch.rand(M, 2 * N, device=dev) * 2 - 1).bfloat16()
            dab = torch.empty_like(ab)
        ts = []
        for r in range(ROUNDS + 1):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            if kind == "plain":
                ops.gemm_nt(a, b, out)
            elif kind == "partial":
                ops.gemm_nt_partial(a, b)
            else:
                ops.gemm_nt_swiglu_bwd(a, b, ab, dab)
            e1.record()
            torch.cuda.synchronize()
            if r:
                ts.append(e0.elapsed_time(e1) * 1e3)
        t = sorted(ts)[len(ts) // 2]
        tot_t += t
        tot_f += 2.0 * M * N * K
        print(f"{name:10s} {kind:8s} {t:8.1f} us {2.0 * M * N * K / t / 1e6:7.0f} TF/s", flush=True)
    print(f"layer total {tot_t:.1f} us, {tot_f / tot_t / 1e6:.0f} TF/s")
