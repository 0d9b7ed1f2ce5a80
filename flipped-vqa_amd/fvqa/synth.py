"""Closed-form synthetic weights and batches for the Flipped-VQA training path.

Nothing here reads a checkpoint, a tokenizer model or a dataset: every tensor is a pure
function of (tensor name, element index), so the same values can be rebuilt bit-for-bit on
the CPU (oracle, fixtures) and on the GPU (parity tests, bench) without shipping weights.

Shapes and names follow the reference state dict (reference llama/model.py:77-85,137-139,
181-182,206-243) and the batch dict produced by dataloader/__init__.py:28-90 and consumed
by llama/model.py:254-264.
"""
from __future__ import annotations

import math
import zlib
from dataclasses import dataclass, field
from typing import Dict, Iterator, Optional, Tuple

import torch

_M32 = 0xFFFFFFFF
_C1 = 0x7FEB352D          # both multipliers < 2**31 so x*c stays below 2**63 for x < 2**32
_C2 = 0x2C1B3C6D
_CHUNK = 1 << 24


def _mix32(x: torch.Tensor) -> torch.Tensor:
    """32-bit avalanche hash carried in int64 lanes (identical on CPU and GPU)."""
    x = x ^ (x >> 16)
    x = (x * _C1) & _M32
    x = x ^ (x >> 15)
    x = (x * _C2) & _M32
    x = x ^ (x >> 16)
    return x


def name_seed(name: str) -> int:
    return zlib.crc32(name.encode("utf-8")) & _M32


def hashed_uniform(name: str, shape, scale: float, device="cpu", offset: float = 0.0) -> torch.Tensor:
    """fp32 tensor, element i = offset + scale * (2*u_i - 1), u_i = hash24(i, name) / 2**24."""
    n = 1
    for s in shape:
        n *= int(s)
    seed = _mix32(torch.tensor([name_seed(name)], dtype=torch.int64, device=device))
    out = torch.empty(n, dtype=torch.float32, device=device)
    for lo in range(0, n, _CHUNK):
        hi = min(n, lo + _CHUNK)
        idx = torch.arange(lo, hi, dtype=torch.int64, device=device)
        h = _mix32(((idx & _M32) + seed) & _M32)
        u = (h >> 8).to(torch.float32) * (1.0 / 16777216.0)
        v = u * 2.0 - 1.0
        v = v * scale
        if offset != 0.0:
            v = v + offset
        out[lo:hi] = v
    return out.reshape(*shape)


@dataclass
class SynthConfig:
    """Model + batch geometry. Defaults = LLaMA-7B, the shape BASELINE.json quotes."""
    dim: int = 4096
    n_heads: int = 32
    n_layers: int = 32
    vocab_size: int = 32000
    multiple_of: int = 256
    norm_eps: float = 1e-6
    adapter_len: int = 10
    adapter_layer: int = 32
    max_feats: int = 10
    max_seq_len: int = 128
    batch_size: int = 8
    bias: float = 3.5
    tau: float = 100.0
    vaq: bool = False
    qav: bool = False
    video_dim: int = 768
    warm: bool = True           # non-zero gate1 so the adapter path carries gradient
    # peaked LM-head logits (parity fixtures that pin the token argmax of the bf16 build on almost every row): the LM
    # head is tied to the token embeddings (x 1/sqrt(dim)) and every temporal embedding is 3 x the embedding of a token,
    # so each position's logits peak at the token (or frame token) that position holds, with a top-2 margin of about
    # half the logit range instead of the few per cent that 32000 independent random rows leave
    peaked: bool = False
    # round 5 (generation fixtures): as `peaked`, but the LM head is tied to the embedding rows through a fixed PERMUTATION pi of
    # the vocabulary — output row pi(t) = embedding row t — so a position holding token t predicts pi(t) != t: a greedy decode
    # walks t -> pi(t) -> pi(pi(t)) ... (31 distinct ids per row, margins as decided as with the identity tie) instead of
    # repeating the token it was fed, and a wrong KV-cache row / position in the token loop changes the next id.
    peaked_perm: bool = False

    @property
    def head_dim(self) -> int:
        return self.dim // self.n_heads

    @property
    def ffn_dim(self) -> int:
        h = int(2 * (4 * self.dim) / 3)
        return self.multiple_of * ((h + self.multiple_of - 1) // self.multiple_of)

    def params_json(self) -> dict:
        return dict(dim=self.dim, multiple_of=self.multiple_of, n_heads=self.n_heads,
                    n_layers=self.n_layers, norm_eps=self.norm_eps, vocab_size=-1)


PRESETS: Dict[str, dict] = {
    # Dh is 128 in every preset: the attention kernels are built for LLaMA's head size.
    "tiny": dict(dim=256, n_heads=2, n_layers=2, vocab_size=512, multiple_of=128,
                 adapter_layer=2, max_seq_len=32, batch_size=2),
    "small": dict(dim=512, n_heads=4, n_layers=3, vocab_size=1024, multiple_of=128,
                  adapter_layer=2, max_seq_len=64, batch_size=3),
    "7b_l2": dict(dim=4096, n_heads=32, n_layers=2, vocab_size=32000, multiple_of=256,
                  adapter_layer=2, max_seq_len=128, batch_size=2),
    "7b": dict(dim=4096, n_heads=32, n_layers=32, vocab_size=32000, multiple_of=256,
               adapter_layer=32, max_seq_len=128, batch_size=8),
    "13b": dict(dim=5120, n_heads=40, n_layers=40, vocab_size=32000, multiple_of=256,
                adapter_layer=40, max_seq_len=128, batch_size=4),
}


def preset(name: str, **over) -> SynthConfig:
    kw = dict(PRESETS[name])
    kw.update(over)
    return SynthConfig(**kw)


def state_spec(cfg: SynthConfig) -> Iterator[Tuple[str, Tuple[int, ...], str]]:
    """(name, shape, kind) for every tensor of the reference state dict on the training path."""
    D, Hf, V = cfg.dim, cfg.ffn_dim, cfg.vocab_size
    yield "tok_embeddings.weight", (V, D), "emb"
    yield "adapter_query.weight", (cfg.adapter_len * cfg.adapter_layer, D), "emb"
    yield "visual_proj.weight", (D, cfg.video_dim), "lin"
    yield "temporal_emb.weight", (cfg.max_feats, D), "tied_temporal" if (cfg.peaked or cfg.peaked_perm) else "emb"
    for i in range(cfg.n_layers):
        p = f"layers.{i}."
        yield p + "attention.wq.weight", (D, D), "lin"
        yield p + "attention.wk.weight", (D, D), "lin"
        yield p + "attention.wv.weight", (D, D), "lin"
        yield p + "attention.wo.weight", (D, D), "lin"
        yield p + "attention.gate1", (1, cfg.n_heads, 1, 1), "gate1"
        yield p + "attention.gate2", (1, cfg.n_heads, 1, 1), "gate2"
        yield p + "feed_forward.w1.weight", (Hf, D), "lin"
        yield p + "feed_forward.w2.weight", (D, Hf), "lin"
        yield p + "feed_forward.w3.weight", (Hf, D), "lin"
        yield p + "attention_norm.weight", (D,), "norm"
        yield p + "ffn_norm.weight", (D,), "norm"
    yield "norm.weight", (D,), "norm"
    yield "output.weight", (V, D), "tied_out_perm" if cfg.peaked_perm else ("tied_out" if cfg.peaked else "lin")


TRAINABLE_MARKS = ("gate", "adapter", "temporal_emb", "visual_proj")   # reference llama_vqa.py:72


def is_trainable(name: str) -> bool:
    return any(m in name for m in TRAINABLE_MARKS)


def make_tensor(cfg: SynthConfig, name: str, shape, kind: str, device="cpu") -> torch.Tensor:
    if kind == "lin":
        return hashed_uniform(name, shape, 1.0 / math.sqrt(shape[-1]), device)
    if kind == "emb":
        return hashed_uniform(name, shape, math.sqrt(3.0), device)
    if kind == "norm":
        return hashed_uniform(name, shape, 0.1, device, offset=1.0)
    if kind in ("tied_out", "tied_out_perm", "tied_temporal"):           # SynthConfig.peaked / peaked_perm
        emb = hashed_uniform("tok_embeddings.weight", (cfg.vocab_size, cfg.dim), math.sqrt(3.0), device)
        if kind == "tied_out":
            return emb * (1.0 / math.sqrt(cfg.dim))
        if kind == "tied_out_perm":
            out = torch.empty_like(emb)
            out[vocab_permutation(cfg.vocab_size, device)] = emb * (1.0 / math.sqrt(cfg.dim))      # row pi(t) <- embedding row t
            return out
        tok = [(97 + 31 * f) % cfg.vocab_size for f in range(shape[0])]
        return 3.0 * emb[torch.tensor(tok, dtype=torch.int64, device=device)]
    if kind == "gate1":
        h = torch.arange(cfg.n_heads, dtype=torch.float32, device=device)
        g = 0.5 * (1.0 - 2.0 * (h % 2)) if cfg.warm else torch.zeros_like(h)
        return g.reshape(shape)
    if kind == "gate2":
        g = hashed_uniform(name, shape, 0.25, device, offset=-cfg.bias) if cfg.warm \
            else torch.full(shape, -cfg.bias, dtype=torch.float32, device=device)
        return g
    raise KeyError(kind)


def vocab_permutation(V: int, device="cpu") -> torch.Tensor:
    """pi(t) = (a t + c) mod V with gcd(a, V) = 1 (a bijection of the vocabulary without short cycles for the sizes used:
    V = 32000 = 2^8 5^3 and V = 512 / 1024 are all coprime with 7919)."""
    a, c = 7919, 12345
    assert math.gcd(a, V) == 1
    return (torch.arange(V, dtype=torch.int64, device=device) * a + c) % V


def state_dict(cfg: SynthConfig, device="cpu") -> Dict[str, torch.Tensor]:
    return {n: make_tensor(cfg, n, s, k, device) for n, s, k in state_spec(cfg)}


# ----------------------------------------------------------------------------------------------
# batches


def _rand_ints(name: str, n: int, lo: int, hi: int) -> torch.Tensor:
    """n integers in [lo, hi) from the same counter hash (CPU)."""
    seed = _mix32(torch.tensor([name_seed(name)], dtype=torch.int64))
    idx = torch.arange(n, dtype=torch.int64)
    h = _mix32((idx + seed) & _M32)
    return lo + (h % max(1, hi - lo))


def make_batch(cfg: SynthConfig, seed: int = 0, batch_size: Optional[int] = None) -> dict:
    """One batch dict with the schema of dataloader/__init__.py:28-90 (CPU tensors).

    Layout mimics NExT-QA prompts: constant prefix of `vs` tokens, `max_feats` frame
    placeholders (id 0), text up to a per-sample length, then pad (id 0). Labels follow
    dataloader/base_dataset.py:63-91: vqa/vaq keep the ids after a prefix index else 0;
    qav is -1 except 0..F-1 on the F frame slots.
    """
    B = batch_size or cfg.batch_size
    S, F, V = cfg.max_seq_len, cfg.max_feats, cfg.vocab_size
    tag = f"batch{seed}"
    vs = max(2, min(19, S // 4 - 1))
    assert vs + F + 8 <= S, "sequence too short for the synthetic prompt layout"
    lo_len = max(vs + F + 7, S - 24)
    lengths = _rand_ints(tag + ".len", B, lo_len, S + 1)

    video = hashed_uniform(tag + ".video", (B, F, cfg.video_dim), math.sqrt(3.0))
    text_id, label = {}, {}
    for task in ("vqa", "vaq", "qav"):
        ids = _rand_ints(f"{tag}.{task}.ids", B * S, 3, V).reshape(B, S)
        lab = torch.zeros(B, S, dtype=torch.int64)
        if task == "qav":
            lab = lab - 1
        for b in range(B):
            ell = int(lengths[b])
            ids[b, ell:] = 0
            if task == "vqa":
                ids[b, vs:vs + F] = 0
                p = max(vs + F + 1, ell - 4)
                lab[b, p:ell] = ids[b, p:ell]
            elif task == "vaq":
                ids[b, vs:vs + F] = 0
                lo = vs + F + 1
                p = lo + int(_rand_ints(f"{tag}.vaq.p{b}", 1, 0, max(1, (ell - lo) // 2))[0])
                lab[b, p:ell] = ids[b, p:ell]
            else:
                pq = ell - F - 1
                ids[b, pq:pq + F] = 0
                lab[b, pq:pq + F] = torch.arange(F)
        text_id[task] = ids.reshape(B, 1, S)
        label[task] = lab.reshape(B, 1, S)
    pq_all = (lengths - F - 1)
    video_index = {
        "vqa": torch.arange(vs, vs + F).repeat(B, 1),
        "vaq": torch.arange(vs, vs + F).repeat(B, 1),
        "qav": pq_all[:, None] + torch.arange(F)[None, :],
    }
    return {
        "vid": [f"synthetic{seed}_{b}" for b in range(B)],
        "video": video,
        "video_len": torch.full((B,), F, dtype=torch.long),
        "text_id": text_id,
        "label": label,
        "video_start": {"vqa": [vs] * B, "vaq": [vs] * B, "qav": [int(x) for x in pq_all]},
        "video_index": video_index,
        "answer": torch.zeros(B, dtype=torch.long),
        "qtype": torch.zeros(B, dtype=torch.long),
    }


class SyntheticLoader:
    """A fixed-length iterable of pre-built batches (stands in for DataLoader in tests/bench)."""

    def __init__(self, cfg: SynthConfig, n_batches: int, rank: int = 0, world: int = 1, pin: bool = False):
        self.batches = [make_batch(cfg, seed=1234 + rank + world * i) for i in range(n_batches)]
        if pin and torch.cuda.is_available():
            for b in self.batches:
                b["video"] = b["video"].pin_memory()
                for k in ("text_id", "label", "video_index"):
                    b[k] = {t: v.pin_memory() for t, v in b[k].items()}

    def __len__(self):
        return len(self.batches)

    def __iter__(self):
        return iter(self.batches)
