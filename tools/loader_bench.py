#!/usr/bin/env python3
"""Batch-producer measurement on the MI355X box: (a) samples/s of the host pipeline alone
(NExT-QA-shaped table -> prompt templates -> labels/masks -> collate -> pinned staging -> H2D), (b) the 7B training
step fed by it against the same step on batches pre-staged in HBM. The vocabulary is a regex stand-in (no LLaMA
tokenizer.model exists offline): token counts per sample match NExT-QA prompts (70-120 tokens)."""
import argparse
import os
import re
import sys
import tempfile
import time
import types
import zlib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "flipped-vqa_amd"))
import pandas as pd  # noqa: E402
import torch  # noqa: E402

import dataloader  # noqa: E402
from fvqa.batch_producer import DeviceBatchProducer  # noqa: E402
from llama.tokenizer import Tokenizer  # noqa: E402

MARK = {"Video": 15167, "Question": 16492, "Answer": 22550, "\n": 13}
PIECE = re.compile(r"\n|\w+|[^\w\s]")


class RegexPieces:
    def encode(self, s):
        return [MARK[p] if p in MARK else 1000 + zlib.crc32(p.encode()) % 14000 for p in PIECE.findall(s)]


def make_table(root, n_rows, n_videos):
    g = torch.Generator().manual_seed(3)
    words = ["why", "what", "how", "where", "did", "the", "man", "dog", "baby", "lady", "after", "before", "ball",
             "run", "sit", "smile", "look", "when", "near", "tree", "garden", "play", "with", "toy", "red"]
    pick = lambda k: " ".join(words[int(i)] for i in torch.randint(0, len(words), (k,), generator=g))
    cols = {"video": [f"v{int(i)}" for i in torch.randint(0, n_videos, (n_rows,), generator=g)],
            "question": [pick(int(torch.randint(5, 14, (1,), generator=g))) for _ in range(n_rows)],
            "answer": torch.randint(0, 5, (n_rows,), generator=g).tolist(),
            "type": ["CW"] * n_rows}
    for i in range(5):
        cols[f"a{i}"] = [pick(int(torch.randint(1, 6, (1,), generator=g))) for _ in range(n_rows)]
    os.makedirs(os.path.join(root, "nextqa", "video_features"))
    for split in ("train", "val"):
        pd.DataFrame(cols).to_csv(os.path.join(root, "nextqa", f"{split}.csv"), index=False)
    feats = {f"v{i}": torch.randn(int(torch.randint(4, 40, (1,), generator=g)), 768, generator=g).half()
             for i in range(n_videos)}
    torch.save(feats, os.path.join(root, "nextqa", "video_features", "clipvitl14.pth"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=4096)
    ap.add_argument("--batch_size", type=int, default=8)
    ap.add_argument("--workers", type=int, default=8)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--model", default="7B")
    ap.add_argument("--n_layers", type=int, default=0)
    ap.add_argument("--sweep", default="", help="comma list of worker counts: repeat the epoch measurement (c) with each")
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    root = tempfile.mkdtemp()
    make_table(root, a.rows, 256)
    args = types.SimpleNamespace(max_feats=10, max_seq_len=128, dataset="nextqa", audio=False, audio_only=False,
                                 audio_merge="none", debug=False, is_generation_task=False, synthetic=True,
                                 data_root=root, batch_size=a.batch_size, num_workers=a.workers, pin_mem=False)
    tok = Tokenizer("/nonexistent/tokenizer.model", args)
    tok.sp_model = RegexPieces()
    loader = dataloader.load_data(args, tok, split="train")

    # (a) host pipeline alone
    prod = DeviceBatchProducer(loader, dev, depth=3)
    t0, n = None, 0
    for i, b in enumerate(prod):
        if i == 8:                                   # workers warmed up
            torch.cuda.synchronize()
            t0, n = time.perf_counter(), 0
        n += b["video"].shape[0]
    torch.cuda.synchronize()
    rate = n / (time.perf_counter() - t0)
    print(f"producer alone: {rate:9.1f} samples/s ({a.workers} workers, batch {a.batch_size}, "
          f"{prod.h2d_bytes / max(1, len(loader)) / 1024:.0f} KiB H2D per batch)", flush=True)

    # (b) the training step fed by it
    from fvqa.optim import FusedAdamW, param_groups_weight_decay
    from llama_vqa import LLaMA_VQA
    from util import misc
    margs = types.SimpleNamespace(
        llama_model_path="/nonexistent/", model=a.model, max_seq_len=128, adapter_len=10, adapter_layer=32,
        max_feats=10, bias=3.5, tau=100.0, vaq=False, qav=False, audio=False, audio_only=False, audio_merge="none",
        debug=False, synthetic=True, random_init=True, dtype="bf16", accum_iter=1, weight_decay=0.14)
    kw = {}
    if a.n_layers:
        kw["n_layers"] = a.n_layers
        margs.adapter_layer = a.n_layers
    model = LLaMA_VQA(margs, **kw).to(dev)
    opt = FusedAdamW(param_groups_weight_decay(model, 0.14), lr=1e-3, betas=(0.9, 0.95), flat=model.flat_params())
    scaler = misc.NativeScalerWithGradNormCount()

    def step(batch):
        opt.zero_grad()
        vqa, vaq, qav = model(batch)
        scaler(vqa + vaq + qav, opt, parameters=None, update_grad=True)

    def timed(batches_iter):
        t0, n = None, 0
        for i, b in enumerate(batches_iter):
            if i == 5:
                torch.cuda.synchronize()
                t0, n = time.perf_counter(), 0
            if i == 5 + a.steps:
                break
            step(b)
            n += b["video"].shape[0]
        torch.cuda.synchronize()
        return n / (time.perf_counter() - t0)

    fed = timed(DeviceBatchProducer(loader, dev, depth=3))
    staged = [b for b, _ in zip(DeviceBatchProducer(loader, dev, depth=3), range(4))]
    staged = [{k: (v.clone() if torch.is_tensor(v) else ({t: (x.clone() if torch.is_tensor(x) else x)
                                                         for t, x in v.items()} if isinstance(v, dict) else v))
               for k, v in b.items()} for b in staged]
    pre = timed(staged[i % 4] for i in range(5 + a.steps))
    print(f"training step ({a.model}{' ' + str(a.n_layers) + ' layers' if a.n_layers else ''}): fed by the producer "
          f"{fed:7.1f} samples/s, batches pre-staged in HBM {pre:7.1f} samples/s ({fed / pre * 100:.1f} %)")

    # (c) the reference's own loop: engine.train_one_epoch (LR schedule, loss read back every iteration, meters)
    import contextlib
    import io
    import engine

    class Limited:                                   # the first n batches of the producer, with a length
        def __init__(self, src, n):
            self.src, self.n = src, n

        def __len__(self):
            return self.n

        def __iter__(self):
            for i, b in enumerate(self.src):
                if i == self.n:
                    return
                yield b

    margs.lr, margs.min_lr, margs.warmup_epochs, margs.epochs = 1e-3, 0.0, 0, 1
    def epoch(n):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        with contextlib.redirect_stdout(io.StringIO()):
            engine.train_one_epoch(model, Limited(DeviceBatchProducer(loader, dev, depth=3), n), opt, 0, scaler,
                                   args=margs)
        torch.cuda.synchronize()
        return time.perf_counter() - t0

    epoch(6)
    t_short, t_long = epoch(a.steps), epoch(3 * a.steps)     # the difference leaves out the start of the loader's workers
    loop = 2 * a.steps * a.batch_size / (t_long - t_short)
    print(f"engine.train_one_epoch fed by the producer: {loop:7.1f} samples/s ({loop / pre * 100:.1f} % of the bare step; "
          f"{a.steps} / {3 * a.steps} iterations took {t_short:.2f} / {t_long:.2f} s; {a.workers} workers, "
          f"{torch.get_num_threads()} torch threads, {len(os.sched_getaffinity(0))} CPUs)", flush=True)
    for w in [int(x) for x in a.sweep.split(",") if x.strip()]:
        args.num_workers = w
        loader = dataloader.load_data(args, tok, split="train")
        epoch(6)
        pre_w = timed(staged[i % 4] for i in range(5 + a.steps))        # the bare step again, right beside it
        t_short, t_long = epoch(a.steps), epoch(3 * a.steps)
        loop = 2 * a.steps * a.batch_size / (t_long - t_short)
        print(f"  sweep: {w} workers: train_one_epoch {loop:7.1f} samples/s = {loop / pre_w * 100:.1f} % of the bare step "
              f"({pre_w:.1f} samples/s measured beside it)", flush=True)


if __name__ == "__main__":
    main()
