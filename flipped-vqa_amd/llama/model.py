"""Drop-in `llama.model` for the Flipped-VQA training path on MI355X.

Same public surface as the reference module (reference llama/model.py:17-29 ModelArgs,
:190-365 Transformer): `Transformer(params, args)`, `forward(data, inference=False) ->
(vqa_loss, vaq_loss, qav_loss)`, the same parameter names/shapes in `state_dict()` — but the
modules here are only parameter containers. No torch arithmetic happens in this file: the
forward/backward is the kernel schedule of fvqa/step.py over libfvqa_hip.so, attached to
autograd through a single Function whose backward writes the trainable gradients.
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Optional

import torch
from torch import nn

from fvqa.step import FlatParams, StepEngine
from llama.tokenizer import Tokenizer


@dataclass
class ModelArgs:  # reference llama/model.py:17-29 (same fields and defaults)
    dim: int = 512
    n_layers: int = 8
    n_heads: int = 8
    vocab_size: int = -1
    multiple_of: int = 256
    norm_eps: float = 1e-5

    max_batch_size: int = 32
    max_seq_len: int = 2048
    adapter_len: int = 10
    adapter_layer: int = 30


def swiglu_hidden(dim: int, multiple_of: int) -> int:
    """SwiGLU width rule of reference llama/model.py:134-135 with hidden_dim = 4*dim (:179)."""
    h = int(2 * (4 * dim) / 3)
    return multiple_of * ((h + multiple_of - 1) // multiple_of)


class _Weight(nn.Module):
    """Holds one `.weight` parameter (stands in for nn.Linear / nn.Embedding / RMSNorm: the
    state-dict key is `<name>.weight`, exactly as in the reference)."""

    def __init__(self, *shape, init: str = "empty"):
        super().__init__()
        w = torch.empty(*shape)
        if init == "ones":
            w.fill_(1.0)
        elif init == "normal":
            w = torch.randn(*shape, dtype=torch.float32).to(w.dtype)
        elif init == "linear":                      # nn.Linear default: U(-1/sqrt(fan_in), 1/sqrt(fan_in))
            b = 1.0 / math.sqrt(shape[-1])
            w = ((torch.rand(*shape, dtype=torch.float32) * 2 - 1) * b).to(w.dtype)
        self.weight = nn.Parameter(w)


class _Attention(nn.Module):   # parameter layout of reference llama/model.py:77-85
    def __init__(self, dim: int, n_heads: int, bias: float):
        super().__init__()
        self.wq, self.wk, self.wv, self.wo = (_Weight(dim, dim) for _ in range(4))
        self.gate1 = nn.Parameter(torch.zeros(1, n_heads, 1, 1))
        self.gate2 = nn.Parameter(torch.ones(1, n_heads, 1, 1) * -bias)


class _FeedForward(nn.Module):  # reference llama/model.py:137-139
    def __init__(self, dim: int, hidden: int):
        super().__init__()
        self.w1, self.w2, self.w3 = _Weight(hidden, dim), _Weight(dim, hidden), _Weight(hidden, dim)


class _Block(nn.Module):        # reference llama/model.py:178-182
    def __init__(self, layer_id: int, p: ModelArgs, bias: float):
        super().__init__()
        self.layer_id = layer_id
        self.attention = _Attention(p.dim, p.n_heads, bias)
        self.feed_forward = _FeedForward(p.dim, swiglu_hidden(p.dim, p.multiple_of))
        self.attention_norm = _Weight(p.dim, init="ones")
        self.ffn_norm = _Weight(p.dim, init="ones")


class _StepFunction(torch.autograd.Function):
    """forward: losses (3,) from the kernel schedule; backward: launches the backward schedule.

    In 'flat' gradient mode (default) the backward accumulates directly into the flat gradient
    buffer that every trainable's .grad views, and returns no per-input gradients. In 'autograd'
    mode it returns ordinary gradient tensors (for third-party wrappers that hook AccumulateGrad,
    e.g. torch DDP)."""

    @staticmethod
    def forward(ctx, model, data, *trainables):
        ctx.model = model
        return model._engine.forward(data)

    @staticmethod
    def backward(ctx, g_losses):
        model = ctx.model
        flat: FlatParams = model._flat
        if model.grad_mode == "flat":
            if not flat.grads_attached():          # optimizer.zero_grad(set_to_none=True) happened
                flat.zero_grad()
                flat.attach_grads()
            model._engine.backward(g_losses.float().contiguous(), flat)
            return (None, None) + tuple(None for _ in flat.offsets)
        keep = flat.flat_grad.clone()
        flat.zero_grad()
        model._engine.backward(g_losses.float().contiguous(), flat)
        out = tuple(flat.grad_view(n).clone() for n in flat.offsets)
        flat.flat_grad.copy_(keep)
        return (None, None) + out


class Transformer(nn.Module):
    def __init__(self, params: ModelArgs, args):
        super().__init__()
        params.max_feats = args.max_feats          # reference llama/model.py:193-194
        params.bias = args.bias
        self.args = args
        self.params = params
        self.vocab_size = params.vocab_size
        self.n_layers = params.n_layers
        self.max_feats = args.max_feats
        if getattr(args, "audio", False):
            raise NotImplementedError("audio fusion variants are outside the MI355X hot path (video-only)")

        self.tokenizer = Tokenizer(model_path=f"{args.llama_model_path}./tokenizer.model", args=args)
        self.eos_id = self.tokenizer.eos_id
        self.answer_token_id = self.tokenizer.a_token_id
        self.q_token_id = self.tokenizer.q_token_id

        self.tok_embeddings = _Weight(params.vocab_size, params.dim)
        self.adapter_query = _Weight(params.adapter_len * params.adapter_layer, params.dim, init="normal")
        self.visual_proj = _Weight(params.dim, 768, init="linear")
        self.temporal_emb = _Weight(self.max_feats, params.dim, init="normal")
        self.adapter_len = params.adapter_len
        self.adapter_layer = params.adapter_layer
        self.layers = nn.ModuleList(_Block(i, params, args.bias) for i in range(params.n_layers))
        self.norm = _Weight(params.dim, init="ones")
        self.output = _Weight(params.vocab_size, params.dim)
        self.tau = args.tau

        self.grad_mode = "flat"
        self._engine: Optional[StepEngine] = None
        self._flat: Optional[FlatParams] = None
        self._engine_key = None

    # ---- helpers used by the step engine ----------------------------------------------------
    def rope_tables(self):
        """cos/sin of reference precompute_freqs_cis (llama/model.py:45-50,245): fp32 angles
        p * 10000^(-2i/Dh) for p < 2*max_seq_len."""
        dh = self.params.dim // self.params.n_heads
        inv = 1.0 / (10000.0 ** (torch.arange(0, dh, 2)[: dh // 2].float() / dh))
        ang = torch.outer(torch.arange(self.params.max_seq_len * 2).float(), inv).float()
        return torch.cos(ang), torch.sin(ang)

    def engine_layer_ids(self):
        return list(range(self.params.n_layers))[-self.adapter_layer:]    # llama/model.py:338

    def gate_views(self, i: int):
        li = self.engine_layer_ids()[i]
        g = self._flat.gates
        return g[li, 0], g[li, 1]

    def trainable_parameters(self):
        return [p for p in self.parameters() if p.requires_grad]

    def ensure_engine(self):
        """Pack weights / flatten trainables on first use (after load_state_dict + .to(device))."""
        w = self.tok_embeddings.weight
        key = (w.data_ptr(), w.dtype, str(w.device), self.adapter_query.weight.data_ptr())
        if self._engine is None or self._engine_key is None or key[:3] != self._engine_key[:3]:
            # fp16 frozen weights — the reference's own storage type (llama_vqa.py:63) — stay fp16: the fp16 build of the kernels
            # (libfvqa_hip_f16.so, v_mfma_f32_16x16x32_f16) takes them as they are. (Rounds 2-4 re-rounded them to bf16 here.)
            self._flat = FlatParams(self)
            self._engine = StepEngine(self)
            w = self.tok_embeddings.weight
            self._engine_key = (w.data_ptr(), w.dtype, str(w.device), self.adapter_query.weight.data_ptr())
        return self._engine

    def flat_params(self) -> FlatParams:
        self.ensure_engine()
        return self._flat

    # ---- reference forward signature --------------------------------------------------------
    @torch.no_grad()
    def inference(self, data):
        """Greedy generation of the answer + nearest-choice matching (reference llama/model.py:367-546):
        -> (most_similar_indices (B,), extracted_answers list of dicts). KV-cached, batched: fvqa/generate.py."""
        from fvqa import generate
        eng = self.ensure_engine()
        ids = generate.greedy_decode(eng, data)
        best, sims, extracted = generate.match_answers(self, data, ids)
        self.last_generation = {"ids": ids, "similarities": sims}
        return best, extracted

    def forward(self, data, inference=False):
        if inference:
            return self.inference(data)
        eng = self.ensure_engine()
        flat = self._flat
        named = dict(self.named_parameters())
        trainables = [named[n] for n in flat.offsets]
        if torch.is_grad_enabled() and any(p.requires_grad for p in trainables):
            losses = _StepFunction.apply(self, data, *trainables)
        else:
            losses = eng.forward(data)
        vqa_loss = losses[0]
        # tensor([0]) of a switched-off loss (llama/model.py:302), made ON the device: torch.tensor([0], device=...) is a
        # pageable host-to-device copy, which blocks the host until the stream reaches it — i.e. until the whole forward
        # has run — and the backward then starts late (C2 step: 29.7 -> 28.7 ms, profiles/r02_gemm_partition_probe.log section 7)
        zero = lambda: torch.zeros(1, dtype=torch.int64, device=losses.device)      # noqa: E731
        vaq_loss = losses[1] if self.args.vaq else zero()
        qav_loss = losses[2] if self.args.qav else zero()
        return vqa_loss, vaq_loss, qav_loss
