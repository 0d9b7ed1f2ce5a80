#!/usr/bin/env python3
"""Generation/eval path measurement on the MI355X box: greedy decode of 31 answer tokens for
a batch of B samples with the KV-cached row-wise decode, against the same decode done the reference's way on
the same kernels (one full forward of the batch per generated token; the reference additionally runs its
forwards one sample at a time)."""
import argparse
import os
import sys
import time
import types

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "flipped-vqa_amd"))
import torch  # noqa: E402

from fvqa import generate, synth  # noqa: E402
from llama_vqa import LLaMA_VQA  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="7B")
    ap.add_argument("--batch_size", type=int, default=8)
    ap.add_argument("--n_layers", type=int, default=0)
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    margs = types.SimpleNamespace(
        llama_model_path="/nonexistent/", model=a.model, max_seq_len=128, adapter_len=10, adapter_layer=32,
        max_feats=10, bias=3.5, tau=100.0, vaq=False, qav=False, audio=False, audio_only=False, audio_merge="none",
        debug=False, synthetic=True, random_init=True, dtype="bf16", accum_iter=1, weight_decay=0.14)
    kw = {}
    if a.n_layers:
        kw["n_layers"] = a.n_layers
        margs.adapter_layer = a.n_layers
    model = LLaMA_VQA(margs, **kw).to(dev).eval()
    p = model.params
    cfg = synth.SynthConfig(dim=p.dim, n_heads=p.n_heads, n_layers=p.n_layers, vocab_size=model.vocab_size,
                            max_seq_len=128, batch_size=a.batch_size, vaq=False, qav=False)
    b = synth.make_batch(cfg, seed=5)
    B = a.batch_size
    b["prefix_index"] = {"vqa": [60 + i for i in range(B)]}
    eng = model.ensure_engine()
    for _ in range(2):
        generate.greedy_decode(eng, b)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 3
    for _ in range(n):
        generate.greedy_decode(eng, b)
    torch.cuda.synchronize()
    t_kv = (time.perf_counter() - t0) / n
    eng.lm_head_rows = "all"        # the reference's re-forward evaluates the head at every position (llama/model.py:439-447)
    with torch.no_grad():
        for _ in range(2):
            eng.forward(b)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(generate.N_NEW):
            eng.forward(b)
        torch.cuda.synchronize()
        t_full = time.perf_counter() - t0
    print(f"{a.model} B={B} S=128: 31 greedy tokens — KV-cached rows {t_kv * 1e3:7.1f} ms/batch "
          f"({B / t_kv:6.1f} samples/s); one full batched forward per token {t_full * 1e3:7.1f} ms/batch "
          f"({B / t_full:6.1f} samples/s); ratio {t_full / t_kv:.2f}x")


if __name__ == "__main__":
    main()
