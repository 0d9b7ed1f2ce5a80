#!/usr/bin/env python3
"""Phase times of the persistent per-token kernel (csrc/decode.hip): one token of LLaMA-7B shapes (random weights), B = 8, S = 128,
stamps of workgroup 0 in the layer before the last (100 MHz counter). Diagnostic; FVQA_DECODE_PREFETCH=0 for the other form."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "flipped-vqa_amd"))
import torch  # noqa: E402
from fvqa import ops  # noqa: E402

dev = "cuda"
B, S, H, Dh, Hf, A, F, L = 8, 128, 32, 128, 11008, 10, 10, int(os.environ.get("DS_LAYERS", "8"))
D = H * Dh
dt = torch.bfloat16
torch.manual_seed(0)
r = lambda *s, sc=1.0: ((torch.rand(*s, device=dev) * 2 - 1) * sc).to(dt)  # noqa: E731
layers = [dict(an=r(D) + 1, wqkv=r(3 * D, D, sc=D ** -0.5), wo=r(D, D, sc=D ** -0.5), fn=r(D) + 1, w13=r(2 * Hf, D, sc=D ** -0.5),
               w2=r(D, Hf, sc=Hf ** -0.5), cache=r(B * S + A, 3 * D), g1=torch.rand(H, device=dev), g2=torch.rand(H, device=dev))
          for _ in range(L)]
table = torch.tensor([[l[k].data_ptr() for k in ("an", "wqkv", "wo", "fn", "w13", "w2", "cache", "g1", "g2")] for l in layers],
                     dtype=torch.int64, device=dev)
ang = torch.outer(torch.arange(2 * S).float(), 1.0 / (10000.0 ** (torch.arange(0, Dh, 2).float() / Dh)))
rope = (torch.cos(ang).to(dev), torch.sin(ang).to(dev))
pos = torch.full((B,), 100, dtype=torch.int64, device=dev)
vs = torch.full((B,), 7, dtype=torch.int32, device=dev)
e = lambda *s: torch.empty(*s, dtype=dt, device=dev)  # noqa: E731
x0, x_out = r(B, D), e(B, D)
scratch = ops.decode_scratch(L, B, H, Dh, Hf, dev)
ws = ops.decode_workspace(dev)


def run():
    ops.decode_token(table, L, x0, x_out, scratch, vs, pos, rope, B, S, H, Dh, Hf, A, F, 1e-5, True, ws)


for _ in range(3):
    run()
torch.cuda.synchronize()
t0 = time.perf_counter()
n = 20
for _ in range(n):
    run()
torch.cuda.synchronize()
dtm = (time.perf_counter() - t0) / n
st = ws[64:64 + 13 * 8].view(torch.int64).cpu().tolist()
names = ["norm1", "qkv strips", "(prefetch) -> barrier", "attention", "barrier", "wo strips", "barrier", "norm2", "w13 strips + swiglu",
         "barrier", "w2 strips", "barrier"]
print(f"{L} layers: {dtm * 1e6:.1f} us per token launch = {dtm * 1e6 / L:.1f} us per layer (weights {405.0:.0f} MB per layer); error word "
      f"{int(ws[16:24].view(torch.int64)[0].item())}")
for i, nme in enumerate(names):
    print(f"  {nme:24s} {(st[i + 1] - st[i]) / 100.0:7.2f} us")
print(f"  layer total (workgroup 0)  {(st[12] - st[0]) / 100.0:7.2f} us")
