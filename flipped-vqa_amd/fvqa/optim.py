"""Optimizer-side host code: timm's weight-decay grouping rule, and AdamW over the flat trainable
buffer driven by the fvqa_adamw_step kernel (replaces torch.optim.AdamW of train.py:120-121)."""
from __future__ import annotations

from typing import List

import torch

from . import ops


def param_groups_weight_decay(model, weight_decay=1e-5, no_weight_decay_list=()):
    """timm.optim.optim_factory.param_groups_weight_decay restated (reference train.py:120; timm is
    an un-vendored dependency): frozen parameters are skipped; 1-D parameters, `.bias` and names in
    the skip list get weight_decay 0; groups are returned [no_decay, decay]."""
    skip = set(no_weight_decay_list)
    decay, no_decay = [], []
    for name, p in model.named_parameters():
        if not p.requires_grad:
            continue
        (no_decay if (p.ndim <= 1 or name.endswith(".bias") or name in skip) else decay).append(p)
    return [{"params": no_decay, "weight_decay": 0.0}, {"params": decay, "weight_decay": weight_decay}]


class FusedAdamW(torch.optim.Optimizer):
    """AdamW (decoupled decay, bias correction, eps outside the sqrt) with torch.optim.AdamW's
    param_groups / state_dict layout. Parameters must live in the model's flat trainable buffer
    (Transformer.flat_params()); adjacent parameters of a group are updated by one kernel launch.
    The update is skipped on device when `found_inf` is set (GradScaler semantics)."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, *, flat):
        # the param-group keys of torch.optim.AdamW (what the reference's optimizer.state_dict() carries and what its
        # load_state_dict expects back): a checkpoint written here loads into the reference's torch AdamW and back
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, amsgrad=False,
                                      maximize=False, foreach=None, capturable=False, differentiable=False,
                                      fused=None, decoupled_weight_decay=True))
        self.flat = flat
        dev = flat.flat.device
        self.exp_avg = torch.zeros_like(flat.flat)
        self.exp_avg_sq = torch.zeros_like(flat.flat)
        self.step_dev = torch.zeros(1, dtype=torch.float32, device=dev)
        self.found_inf = torch.zeros(1, dtype=torch.float32, device=dev)
        self.grad_sync = None                    # set by the data-parallel wrapper
        base = flat.flat.data_ptr()
        self._ranges: List[List[tuple]] = []
        # parameters that never receive a gradient (gates of layers the engine skips, llama/model.py:338): torch
        # AdamW leaves a parameter whose .grad is None untouched — no decay, no moments — so they are not stepped
        idle = set(flat.idle_offsets()) if hasattr(flat, "idle_offsets") else set()
        for g in self.param_groups:
            spans = []
            for p in g["params"]:
                off = (p.data_ptr() - base) // 4
                if not (0 <= off and off + p.numel() <= flat.flat.numel()) or p.dtype != torch.float32:
                    raise ValueError("FusedAdamW: parameter is not a view of the flat trainable buffer")
                if off not in idle:
                    spans.append((off, off + p.numel()))
                self.state[p] = {"step": self.step_dev, "exp_avg": self.exp_avg[off:off + p.numel()].view(p.shape),
                                 "exp_avg_sq": self.exp_avg_sq[off:off + p.numel()].view(p.shape)}
            spans.sort()
            merged = []
            for lo, hi in spans:
                if merged and merged[-1][1] == lo:
                    merged[-1] = (merged[-1][0], hi)
                else:
                    merged.append((lo, hi))
            self._ranges.append(merged)

    def zero_grad(self, set_to_none: bool = True):
        # one memset of the flat gradient buffer; .grad views stay attached
        self.flat.zero_grad()
        if not self.flat.grads_attached():
            self.flat.attach_grads()

    @torch.no_grad()
    def step(self, closure=None, found_inf=None):
        fi = found_inf if found_inf is not None else None
        f = self.flat
        for g, ranges in zip(self.param_groups, self._ranges):
            b1, b2 = g["betas"]
            for lo, hi in ranges:
                ops.adamw_step(f.flat[lo:hi], f.flat_grad[lo:hi], self.exp_avg[lo:hi], self.exp_avg_sq[lo:hi],
                               g["lr"], b1, b2, g["eps"], g["weight_decay"], self.step_dev, fi)
        if found_inf is None:                    # stand-alone use: advance the step here
            self.found_inf.zero_()
            ops.scaler_update(self.step_dev, None, None, self.found_inf)
        return None

    def state_dict(self):
        sd = super().state_dict()
        for st in sd["state"].values():
            st["step"] = st["step"].detach().clone().reshape(())
        return sd

    def load_state_dict(self, state_dict):
        steps = [float(s["step"]) for s in state_dict["state"].values()]
        ids = [i for g in state_dict["param_groups"] for i in g["params"]]
        mine = [p for g in self.param_groups for p in g["params"]]
        for i, p in zip(ids, mine):
            s = state_dict["state"].get(i)
            if s is None:
                continue
            self.state[p]["exp_avg"].copy_(s["exp_avg"])
            self.state[p]["exp_avg_sq"].copy_(s["exp_avg_sq"])
        if steps:
            self.step_dev.fill_(steps[0])
        for g, sg in zip(self.param_groups, state_dict["param_groups"]):
            for k, v in sg.items():
                if k != "params":
                    g[k] = v
