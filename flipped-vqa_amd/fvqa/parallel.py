"""Data-parallel training over one 8xMI355X node: replicated model, samples sharded by rank,
ONE exchange step — an all-reduce(mean) of the flat fp32 trainable-gradient buffer (18 MB for 7B)
over RCCL/xGMI per optimizer step. Replaces torch DDP of reference train.py:115-117, whose bucketed
reducer all-reduces the same ~4.5 M scalars on every backward.

The averaging rule is backend-agnostic (tests drive it with gloo on CPU tensors)."""
from __future__ import annotations

import os

import torch
import torch.distributed as dist


def allreduce_sum_(flat_grad: torch.Tensor, group=None) -> int:
    """In-place SUM over ranks of one flat gradient buffer; returns the number of replicas summed (the divisor of the
    mean, which the fused unscale kernel applies together with 1/loss-scale: no pass of its own)."""
    if not (dist.is_available() and dist.is_initialized()):
        return 1
    world = dist.get_world_size(group)
    # (FVQA_DP_FORCE_ALLREDUCE=1: tests run the collective on a one-rank group too — RCCL beside libfvqa_hip.so)
    if world > 1 or os.environ.get("FVQA_DP_FORCE_ALLREDUCE") == "1":
        dist.all_reduce(flat_grad, op=dist.ReduceOp.SUM, group=group)
    return world


def allreduce_mean_(flat_grad: torch.Tensor, group=None) -> torch.Tensor:
    """In-place mean over ranks of one flat gradient buffer (what DDP computes per bucket)."""
    if not (dist.is_available() and dist.is_initialized()):
        return flat_grad
    world = dist.get_world_size(group)
    if world == 1:
        return flat_grad
    dist.all_reduce(flat_grad, op=dist.ReduceOp.SUM, group=group)
    flat_grad.div_(world)
    return flat_grad


def shard_indices(n_items: int, rank: int, world: int, epoch: int = 0, shuffle: bool = True, seed: int = 0):
    """DistributedSampler's rule (reference dataloader/__init__.py:19-21): permute with seed+epoch,
    pad by wrapping to a multiple of world, take rank::world."""
    if shuffle:
        g = torch.Generator()
        g.manual_seed(seed + epoch)
        idx = torch.randperm(n_items, generator=g).tolist()
    else:
        idx = list(range(n_items))
    total = ((n_items + world - 1) // world) * world
    idx += idx[: total - len(idx)]
    return idx[rank:total:world]


class DataParallel(torch.nn.Module):
    """Minimal DDP stand-in: exposes `.module`, forwards calls, and synchronises gradients through
    `sync_grads()` (invoked by the loss scaler right before unscale/step on accumulation
    boundaries — mathematically identical to DDP's every-backward all-reduce).

    Like torch DDP at construction (reference train.py:115-117; DDP broadcasts rank 0's module state in its
    constructor), rank 0's trainable parameters are broadcast to every rank: train.py seeds each rank with
    seed+rank (train.py:87), so the randomly initialised adapter / projection / temporal parameters would
    otherwise differ between replicas and the averaged gradient would be applied to different models."""

    def __init__(self, module, group=None):
        super().__init__()
        self.module = module
        self.group = group
        self.comm_events = None
        self.last_sync_ordering = None
        self.broadcast_params()

    def forward(self, *a, **k):
        return self.module(*a, **k)

    def _bcast(self, t: torch.Tensor):
        if dist.is_available() and dist.is_initialized() and dist.get_world_size(self.group) > 1:
            src = dist.get_global_rank(self.group, 0) if self.group is not None else 0
            dist.broadcast(t, src=src, group=self.group)

    def broadcast_params(self):
        """rank 0's flat trainable buffer -> every rank (one 18 MB broadcast for 7B)."""
        self._bcast(self.module.flat_params().flat)

    def broadcast_optimizer(self, optimizer, loss_scaler=None):
        """After a resume: rank 0's AdamW moments / step counter (and loss-scale state) -> every rank, so
        that replicas stay bitwise identical whatever each rank read from disk."""
        for t in (optimizer.exp_avg, optimizer.exp_avg_sq, optimizer.step_dev):
            self._bcast(t)
        if loss_scaler is not None and getattr(loss_scaler, "_dev", None) is not None:
            self._bcast(loss_scaler._scale)
            self._bcast(loss_scaler._tracker)

    def sync_grads(self) -> int:
        """ONE all-reduce(SUM) of the flat gradient buffer over RCCL/xGMI; returns the replica count, by which the
        loss scaler's unscale kernel divides (fvqa_grad_unscale_norm grad_div). With `comm_events` set to a list,
        each call appends a (start, end) event pair recorded on the current stream (bench.py: allreduce_ms)."""
        flat = self.module.flat_params()
        # the error lane behind the gradients rides in the same collective: 1 on a rank whose persistent-GEMM error word is
        # raised (a timed-out split-K exchange), so that every rank sees a non-zero sum, skips this step and stops with the
        # others instead of applying an update the faulty rank does not apply
        buf = getattr(flat, "grad_store", None)
        if buf is None:
            buf = flat.flat_grad
        else:
            word = self.error_word()
            if word is not None:
                flat.err_lane.copy_(word.view(torch.int64) != 0)
            else:
                flat.err_lane.zero_()
        # Stream ordering made explicit (torch orders a collective behind the CURRENT stream only: ProcessGroupNCCL makes its
        # own stream wait for an event of the current one, and the blocking form makes the current stream wait for the
        # collective): when the backward produced the gradients on another stream than the one the optimizer step runs on,
        # the current stream first waits for the producer — so the all-reduce, and the unscale kernel queued behind it on the
        # current stream, see finished gradients. `last_sync_ordering` records which case the last call was (tests).
        self.last_sync_ordering = "host"
        if flat.flat_grad.is_cuda:
            cur = torch.cuda.current_stream(flat.flat_grad.device)
            prod = getattr(flat, "producer_stream", None)
            if prod is not None and prod != cur:
                cur.wait_stream(prod)
                self.last_sync_ordering = "waited_for_producer_stream"
            else:
                self.last_sync_ordering = "same_stream"
        ev = self.comm_events
        if ev is not None and flat.flat_grad.is_cuda:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            n = allreduce_sum_(buf, self.group)
            e1.record()
            ev.append((e0, e1))
            return n
        return allreduce_sum_(buf, self.group)

    def error_word(self):
        """8-byte device view of this rank's persistent-GEMM error word (None before the first such launch)."""
        fg = self.module.flat_params().flat_grad
        if not fg.is_cuda:
            return None
        from . import ops
        return ops.gemm_error_word(fg.device)
