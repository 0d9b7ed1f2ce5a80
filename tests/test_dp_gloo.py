"""The data-parallel exchange step on CPU: 2 ranks over gloo. Each rank runs the oracle's
forward/backward on ITS shard of the batches, the flat gradient buffers are averaged with the same
fvqa.parallel.allreduce_mean_ the RCCL path uses, and the result must equal the average of the two
per-shard gradients computed in one process (DDP's rule, reference train.py:115-117)."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from fvqa import synth
from fvqa.parallel import allreduce_mean_, shard_indices
from oracle import ref_cpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _flat(grads):
    return torch.cat([grads[n].reshape(-1) for n in sorted(grads)])


def _worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    cfg = synth.preset("tiny", vaq=True, qav=True)
    model = ref_cpu.RefModel(cfg, synth.state_dict(cfg), dtype=torch.float64)
    mine = shard_indices(4, rank, world, shuffle=False)          # batches 0..3 -> rank-strided
    flat = torch.zeros_like(_flat(model.step(synth.make_batch(cfg, seed=mine[0]))["grads"]))
    for s in mine:                                               # accumulate over the local shard
        flat += _flat(model.step(synth.make_batch(cfg, seed=s))["grads"])
    allreduce_mean_(flat)
    torch.save(flat, os.path.join(out_dir, f"r{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_flat_gradient_allreduce_mean_two_ranks(tmp_path):
    world, port = 2, _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    r0, r1 = torch.load(tmp_path / "r0.pt"), torch.load(tmp_path / "r1.pt")
    assert torch.equal(r0, r1)                                   # every rank ends with the same buffer
    cfg = synth.preset("tiny", vaq=True, qav=True)
    model = ref_cpu.RefModel(cfg, synth.state_dict(cfg), dtype=torch.float64)
    per_rank = []
    for rank in range(world):
        acc = None
        for s in shard_indices(4, rank, world, shuffle=False):
            g = _flat(model.step(synth.make_batch(cfg, seed=s))["grads"])
            acc = g if acc is None else acc + g
        per_rank.append(acc)
    want = (per_rank[0] + per_rank[1]) / world
    assert torch.allclose(r0, want, rtol=1e-12, atol=1e-14)


def test_allreduce_is_identity_without_process_group():
    t = torch.arange(5.0)
    assert allreduce_mean_(t.clone()).equal(t)
