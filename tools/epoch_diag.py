#!/usr/bin/env python3
"""Where does engine.train_one_epoch lose time against the bare step? (tuning aid) Same model, batches resident in HBM vs fed
by the batch producer; rates from the difference of a long and a short epoch."""
import contextlib, io, os, sys, time, types
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "flipped-vqa_amd")); sys.path.insert(0, ROOT)
os.environ.setdefault("FVQA_SYNTHETIC_TOKENIZER", "1")
import torch
import engine
import util.misc as misc
from fvqa import synth
from fvqa.optim import FusedAdamW, param_groups_weight_decay
from llama_vqa import LLaMA_VQA
dev = torch.device("cuda", 0)
args = types.SimpleNamespace(llama_model_path="/nonexistent/", model="7B", max_seq_len=128, adapter_len=10, adapter_layer=32,
    max_feats=10, bias=3.5, tau=100.0, vaq=False, qav=False, audio=False, audio_only=False, audio_merge="none", debug=False,
    synthetic=True, random_init=True, dtype="bf16", accum_iter=1, weight_decay=0.14, lr=1e-3, min_lr=0.0, warmup_epochs=0, epochs=1)
model = LLaMA_VQA(args).to(dev)
opt = FusedAdamW(param_groups_weight_decay(model, 0.14), lr=1e-3, betas=(0.9, 0.95), flat=model.flat_params())
scaler = misc.NativeScalerWithGradNormCount()
p = model.params
cfg = synth.SynthConfig(dim=p.dim, n_heads=p.n_heads, n_layers=p.n_layers, vocab_size=model.vocab_size, max_seq_len=128, batch_size=8)
res = []
for i in range(4):
    b = synth.make_batch(cfg, seed=i); b["video"] = b["video"].to(dev)
    for k in ("text_id", "label", "video_index"): b[k] = {t: v.to(dev) for t, v in b[k].items()}
    res.append(b)
class L:
    def __init__(s, n): s.n = n
    def __len__(s): return s.n
    def __iter__(s): return (res[i % 4] for i in range(s.n))
def epoch(loader):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    with contextlib.redirect_stdout(io.StringIO()):
        engine.train_one_epoch(model, loader, opt, 0, scaler, args=args)
    torch.cuda.synchronize(); return time.perf_counter() - t0
def bare(n):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(n):
        opt.zero_grad(); a, b, c = model(res[i % 4]); scaler(a + b + c, opt, parameters=None, update_grad=True)
    torch.cuda.synchronize(); return time.perf_counter() - t0
epoch(L(6)); bare(6)
for rep in range(2):
    ts, tl = epoch(L(40)), epoch(L(120))
    bs, bl = bare(40), bare(120)
    print(f"resident batches: train_one_epoch {(tl - ts) / 80 * 1e3:.3f} ms/iter, bare step {(bl - bs) / 80 * 1e3:.3f} ms/iter")
