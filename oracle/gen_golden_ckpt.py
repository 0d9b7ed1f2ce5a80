#!/usr/bin/env python3
"""TEST INFRASTRUCTURE — checkpoint-format fixtures written by the REFERENCE itself.

Runs ONLY in the build container (needs /root/reference). Two on-disk formats sit either side of the hot path:

  1. Meta's model-parallel shards `consolidated.NN.pth`, merged into one replica by the loop INSIDE the reference's
     `llama_vqa.LLaMA_VQA` (llama_vqa.py:24-58). That function is run as it stands on a synthetic 2-shard checkpoint
     (tiny dims, fp16 like Meta's files, plus the `rope.freqs` buffer real shards carry); the Transformer it would
     build is replaced by a recorder whose `load_state_dict` captures the merged dictionary (shims: stub tokenizer,
     `torch.set_default_tensor_type` no-op — no CUDA here). -> tests/golden/ckpt_merge.npz: every shard tensor and
     every merged tensor, key order included.
  2. The trainable-only `checkpoint_best.pth` of `util.misc.save_model` (util/misc.py:297-320): the reference model
     (fp32-shim, closed-form weights, as oracle/gen_golden.py builds it) takes ONE optimizer step through the
     reference's own `NativeScalerWithGradNormCount` (its `torch.cuda.amp.GradScaler` swapped for the CPU GradScaler
     of the same class so that it is enabled) with `torch.optim.AdamW(betas=(0.9, 0.95))` over timm's parameter groups
     (timm is absent: the restated rule of fvqa.optim), then `misc.save_model(...)` writes the file.
     -> tests/golden/ckpt_reference.pth (the file itself: data written by the reference) and ckpt_reference.npz
     (the same content as plain arrays + the key / dtype lists the product's own writer must reproduce).

usage: python oracle/gen_golden_ckpt.py
"""
import os
import sys
import tempfile
import types
import json

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "flipped-vqa_amd"))
sys.path.insert(0, ROOT)
from fvqa import synth  # noqa: E402
from fvqa.optim import param_groups_weight_decay  # noqa: E402
from oracle.gen_golden import REF, build_reference, install_shims  # noqa: E402

GOLDEN = os.path.join(ROOT, "tests", "golden")
CKPT_CFG = dict(dim=128, n_heads=1, n_layers=2, vocab_size=64, multiple_of=32, adapter_layer=2, max_seq_len=24,
                batch_size=2, vaq=True, qav=True)


def shard_tensors(params, world):
    """Deterministic fp16 shards with Meta's shapes: column-parallel on dim 0, row-parallel (wo, w2, tok_embeddings)
    on dim 1, norms replicated, plus rope.freqs."""
    D, L, V = params["dim"], params["n_layers"], 48
    Hf = 96
    g = torch.Generator().manual_seed(1234)
    rnd = lambda *s: torch.randn(*s, generator=g).to(torch.float16)     # noqa: E731  (.half() is shimmed to fp32 here)
    shards = []
    for _ in range(world):
        sd = {"tok_embeddings.weight": rnd(V, D // world), "norm.weight": None, "output.weight": rnd(V // world, D)}
        for i in range(L):
            p = f"layers.{i}."
            for n in ("attention.wq.weight", "attention.wk.weight", "attention.wv.weight"):
                sd[p + n] = rnd(D // world, D)
            sd[p + "attention.wo.weight"] = rnd(D, D // world)
            sd[p + "feed_forward.w1.weight"] = rnd(Hf // world, D)
            sd[p + "feed_forward.w2.weight"] = rnd(D, Hf // world)
            sd[p + "feed_forward.w3.weight"] = rnd(Hf // world, D)
            sd[p + "attention_norm.weight"] = None
            sd[p + "ffn_norm.weight"] = None
        sd["rope.freqs"] = None
        shards.append(sd)
    for k in shards[0]:                                       # replicated tensors: the same values in every shard
        if shards[0][k] is None:
            t = rnd(8) if k == "rope.freqs" else rnd(D)
            for sd in shards:
                sd[k] = t.clone()
    return shards


def gen_merge(M):
    import llama_vqa as R                                     # the reference's module (sys.path has /root/reference)
    assert R.__file__.startswith(REF)
    params = dict(dim=32, multiple_of=32, n_heads=2, n_layers=2, norm_eps=1e-6, vocab_size=-1)
    shards = shard_tensors(params, 2)
    captured = {}

    class Recorder:
        def __init__(self, model_args, args):
            pass

        def load_state_dict(self, sd, strict=True):
            captured["sd"] = sd
            return [], []

        def named_parameters(self):
            return []

    class Tok:
        def __init__(self, model_path=None, args=None):
            self.n_words = 64

    with tempfile.TemporaryDirectory() as d:
        os.makedirs(os.path.join(d, "7B"))
        with open(os.path.join(d, "7B", "params.json"), "w") as f:
            json.dump(params, f)
        for r, sd in enumerate(shards):
            torch.save(sd, os.path.join(d, "7B", f"consolidated.{r:02d}.pth"))
        keep = (R.Transformer, R.Tokenizer, torch.set_default_tensor_type)
        R.Transformer, R.Tokenizer, torch.set_default_tensor_type = Recorder, Tok, (lambda *a, **k: None)
        try:
            R.LLaMA_VQA(types.SimpleNamespace(llama_model_path=d + "/", model="7B", max_seq_len=16, adapter_len=2,
                                              adapter_layer=2))
        finally:
            R.Transformer, R.Tokenizer, torch.set_default_tensor_type = keep
    merged = captured["sd"]
    out = {"merged_keys": np.array(list(merged.keys())), "shard_keys": np.array(list(shards[0].keys())),
           "n_layers": np.int32(params["n_layers"])}
    for r, sd in enumerate(shards):
        for k, t in sd.items():
            out[f"shard{r}__{k}"] = t.numpy()
    for k, t in merged.items():
        out[f"merged__{k}"] = t.numpy()
    path = os.path.join(GOLDEN, "ckpt_merge.npz")
    np.savez_compressed(path, **out)
    print(f"[ckpt_merge] {len(merged)} merged tensors -> {path} ({os.path.getsize(path) / 1024:.0f} KiB)")


def gen_checkpoint(M):
    # the reference's util/misc.py, loaded from where it lies (its `util` has no __init__.py, so a plain
    # `import util.misc` would resolve to the product's package of the same name)
    import importlib.util
    spec = importlib.util.spec_from_file_location("reference_util_misc", os.path.join(REF, "util", "misc.py"))
    rmisc = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(rmisc)
    assert rmisc.__file__.startswith(REF)
    cfg = synth.SynthConfig(**CKPT_CFG)
    model, args = build_reference(M, cfg)
    model.train(True)
    torch.cuda.amp.GradScaler = lambda *a, **k: torch.amp.GradScaler("cpu")   # same class, enabled without CUDA
    scaler = rmisc.NativeScalerWithGradNormCount()
    opt = torch.optim.AdamW(param_groups_weight_decay(model, 0.14), lr=0.01, betas=(0.9, 0.95))
    batch = synth.make_batch(cfg, seed=0)
    opt.zero_grad()
    vqa, vaq, qav = model(batch)
    scaler(vqa + vaq + qav, opt, parameters=model.parameters(), update_grad=True)
    args.output_dir = GOLDEN
    args.lr, args.weight_decay = 0.01, 0.14
    rmisc.save_model(args=args, epoch=3, model=model, model_without_ddp=model, optimizer=opt, loss_scaler=scaler,
                     name="ckpt_reference")
    path = os.path.join(GOLDEN, "ckpt_reference.pth")
    ck = torch.load(path, map_location="cpu", weights_only=False)
    out = {"top_keys": np.array(list(ck.keys())), "model_keys": np.array(list(ck["model"].keys())),
           "epoch": np.int32(ck["epoch"]), "scaler_json": np.array(json.dumps(ck["scaler"])),
           "losses": np.array([vqa.item(), vaq.item(), qav.item()], dtype=np.float32)}
    for k, t in ck["model"].items():
        out[f"model__{k}"] = t.detach().numpy()
    osd = ck["optimizer"]
    groups = [{k: (list(v) if isinstance(v, (list, tuple)) else v) for k, v in g.items()} for g in osd["param_groups"]]
    out["param_groups_json"] = np.array(json.dumps(groups))
    out["state_ids"] = np.array(sorted(osd["state"].keys()), dtype=np.int64)
    for i, st in osd["state"].items():
        out[f"state_keys__{i}"] = np.array(list(st.keys()))
        for k, v in st.items():
            out[f"state__{i}__{k}"] = v.detach().numpy() if torch.is_tensor(v) else np.float32(v)
    # optimizer param index -> parameter name (timm rule: [no_decay, decay] groups, named_parameters order inside)
    names = [n for n, p in model.named_parameters() if p.requires_grad]
    no_decay = [n for n in names if dict(model.named_parameters())[n].ndim <= 1 or n.endswith(".bias")]
    decay = [n for n in names if n not in no_decay]
    out["opt_param_names"] = np.array(no_decay + decay)
    # the reference trains on: ONE more step (batch seed 1) from the state it just saved — what a run resumed from
    # that file must reproduce
    opt.zero_grad()
    v2, a2, q2 = model(synth.make_batch(cfg, seed=1))
    scaler(v2 + a2 + q2, opt, parameters=model.parameters(), update_grad=True)
    out["losses_step2"] = np.array([v2.item(), a2.item(), q2.item()], dtype=np.float32)
    for n, p_ in model.named_parameters():
        if p_.requires_grad:
            out[f"after2__{n}"] = p_.detach().numpy().copy()
    out["scaler_after2_json"] = np.array(json.dumps(scaler.state_dict()))
    np.savez_compressed(os.path.join(GOLDEN, "ckpt_reference.npz"), **out)
    print(f"[ckpt_reference] losses {vqa.item():.6f} {vaq.item():.6f} {qav.item():.6f}; {len(ck['model'])} trainables; "
          f"scaler {ck['scaler']}; {path} ({os.path.getsize(path) / 1024:.0f} KiB)")


def main():
    M = install_shims()
    gen_merge(M)
    gen_checkpoint(M)


if __name__ == "__main__":
    main()
