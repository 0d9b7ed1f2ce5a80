"""Model factory with the reference's entry point `LLaMA_VQA(args) -> Transformer`
(reference llama_vqa.py:6-78): params.json -> ModelArgs, merge 1..N Meta checkpoint shards,
strict=False load, then the freeze policy (gate / adapter / temporal_emb / visual_proj train in
fp32, everything else frozen).

MI355X specifics: frozen weights are stored in bf16 (`args.dtype`, 'fp32' selects the exact-fp32
validation kernels) and are materialised directly on the GPU; `args.random_init` (or no shard on
disk + FVQA_RANDOM_INIT=1) fills them with the closed-form generator of fvqa.synth instead of
reading a checkpoint, which is how bench.py and the parity tests build a 7B-shaped model offline.
"""
import json
import os
from pathlib import Path

import torch

from fvqa import synth
from llama import ModelArgs, Tokenizer, Transformer

# tensor-parallel split dimension of each Meta shard tensor (-1: replicated)
_SPLIT_DIM = {"tok_embeddings.weight": 1, "norm.weight": -1, "output.weight": 0,
              "attention_norm.weight": -1, "ffn_norm.weight": -1,
              "attention.wq.weight": 0, "attention.wk.weight": 0, "attention.wv.weight": 0,
              "feed_forward.w1.weight": 0, "feed_forward.w3.weight": 0,
              "attention.wo.weight": 1, "feed_forward.w2.weight": 1}

_KNOWN = {"7B": dict(dim=4096, multiple_of=256, n_heads=32, n_layers=32, norm_eps=1e-6, vocab_size=-1),
          "13B": dict(dim=5120, multiple_of=256, n_heads=40, n_layers=40, norm_eps=1e-6, vocab_size=-1)}


def merge_shards(shards, n_layers):
    """One replica from Meta's model-parallel shards: column-parallel tensors concatenate on dim 0,
    row-parallel (wo, w2) and tok_embeddings on dim 1, norms are replicated."""
    if len(shards) == 1:
        return shards[0]
    full = {}
    for name in shards[0]:
        short = name.split(".", 2)[2] if name.startswith("layers.") else name
        dim = _SPLIT_DIM.get(short)
        if dim is None:
            continue                      # e.g. rope.freqs: recomputed, never loaded
        full[name] = shards[0][name].clone() if dim < 0 else torch.cat([s[name] for s in shards], dim=dim)
    return full


def _storage_dtype(args):
    name = str(getattr(args, "dtype", "bf16")).lower()
    # "fp16": the reference's own storage type (llama_vqa.py:63 builds under HalfTensor); the module holds the shards' fp16 values
    # exactly and the fp16 build of the kernels (libfvqa_hip_f16.so) computes on them as they are
    return {"bf16": torch.bfloat16, "bfloat16": torch.bfloat16, "fp32": torch.float32, "float32": torch.float32,
            "fp16": torch.float16, "float16": torch.float16}[name]


def LLaMA_VQA(args, **kwargs):
    model_dir = Path(f"{args.llama_model_path}{args.model}")
    random_init = bool(getattr(args, "random_init", False)) or os.environ.get("FVQA_RANDOM_INIT") == "1"
    pj = model_dir / "params.json"
    if pj.is_file():
        params = json.loads(pj.read_text())
    elif random_init and str(args.model).upper().replace("LLAMA", "").split("_")[0] in _KNOWN:
        params = dict(_KNOWN[str(args.model).upper().replace("LLAMA", "").split("_")[0]])
    else:
        raise FileNotFoundError(pj)
    for k, v in kwargs.items():           # e.g. n_layers=2 for reduced-depth parity runs
        params[k] = v
    tokenizer = Tokenizer(model_path=f"{args.llama_model_path}/tokenizer.model", args=args)
    print(f"Using model: {args.model}")

    shards = []
    for ck in sorted(model_dir.glob("*.pth")) if model_dir.is_dir() else []:
        print("loading from", ck)
        shards.append(torch.load(ck, map_location="cpu"))
    if not shards and not random_init:
        raise FileNotFoundError(f"no *.pth checkpoint under {model_dir} (set args.random_init for synthetic weights)")

    model_args = ModelArgs(max_seq_len=args.max_seq_len, max_batch_size=32, adapter_len=args.adapter_len,
                           adapter_layer=args.adapter_layer, **params)
    model_args.vocab_size = tokenizer.n_words
    device = torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() else torch.device("cpu")
    prev = torch.get_default_dtype()
    torch.set_default_dtype(_storage_dtype(args))
    try:
        with torch.device(device):
            model = Transformer(model_args, args)
    finally:
        torch.set_default_dtype(prev)

    if shards:
        model.load_state_dict(merge_shards(shards, params["n_layers"]), strict=False)
    for name, p in model.named_parameters():     # freeze policy
        p.requires_grad = synth.is_trainable(name)
        if p.requires_grad:
            p.data = p.data.float()
    if not shards:
        fill_closed_form(model)                  # after the fp32 cast of the trainables
    return model


@torch.no_grad()
def fill_closed_form(model, cfg=None):
    """Overwrite every parameter with fvqa.synth's closed-form value (generated on the parameter's
    own device, cast to its dtype)."""
    p = model.params
    cfg = cfg or synth.SynthConfig(dim=p.dim, n_heads=p.n_heads, n_layers=p.n_layers, vocab_size=model.vocab_size,
                                   multiple_of=p.multiple_of, norm_eps=p.norm_eps, adapter_len=p.adapter_len,
                                   adapter_layer=p.adapter_layer, max_feats=model.max_feats,
                                   max_seq_len=p.max_seq_len, bias=model.args.bias, tau=model.args.tau)
    own = dict(model.named_parameters())
    for name, shape, kind in synth.state_spec(cfg):
        t = own[name]
        t.copy_(synth.make_tensor(cfg, name, shape, kind, device=t.device).to(t.dtype))
    return model
