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


class _Flat:
    def __init__(self, rank):
        g = torch.Generator().manual_seed(50 + rank)
        self.flat = torch.randn(1000, generator=g)
        self.flat_grad = torch.randn(1000, generator=g)


class _Module(torch.nn.Module):
    def __init__(self, rank):
        super().__init__()
        self._f = _Flat(rank)

    def flat_params(self):
        return self._f


class _Opt:
    def __init__(self, rank):
        self.exp_avg = torch.full((1000,), float(rank))
        self.exp_avg_sq = torch.full((1000,), 2.0 * rank)
        self.step_dev = torch.full((1,), 3.0 + rank)


def _dp_worker(rank, world, port, out_dir):
    from fvqa.parallel import DataParallel
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    m = _Module(rank)
    g_local = m.flat_params().flat_grad.clone()
    net = DataParallel(m)                           # constructor broadcasts rank 0's parameters (as torch DDP does)
    n_rep = net.sync_grads()                        # all-reduce(SUM); returns the divisor the unscale kernel applies
    opt = _Opt(rank)
    net.broadcast_optimizer(opt)
    torch.save(dict(flat=m.flat_params().flat, grad=m.flat_params().flat_grad, g_local=g_local, n_rep=n_rep, m=opt.exp_avg,
                    v=opt.exp_avg_sq, step=opt.step_dev), os.path.join(out_dir, f"dp{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_dataparallel_broadcasts_rank0_state_and_averages_grads(tmp_path):
    """Replicas seeded seed+rank (reference train.py:87) must still start from ONE set of trainables: DDP broadcasts
    rank 0's at construction (train.py:115-117); so does fvqa.parallel.DataParallel."""
    world, port = 2, _free_port()
    mp.spawn(_dp_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    r0, r1 = torch.load(tmp_path / "dp0.pt"), torch.load(tmp_path / "dp1.pt")
    assert torch.equal(r0["flat"], _Flat(0).flat) and torch.equal(r1["flat"], r0["flat"])
    assert not torch.equal(_Flat(1).flat, _Flat(0).flat)
    # sync_grads leaves the SUM and hands back the replica count (fvqa_grad_unscale_norm's grad_div: the mean rides in
    # the unscale kernel's factor 1/(scale*world) — exact for the power-of-two world sizes of one node)
    want = r0["g_local"] + r1["g_local"]
    assert r0["n_rep"] == world and r1["n_rep"] == world
    assert torch.equal(r0["grad"], want) and torch.equal(r1["grad"], want)
    for k, v in (("m", 0.0), ("v", 0.0), ("step", 3.0)):
        assert torch.all(r0[k] == v) and torch.all(r1[k] == v)


class _FlatLane(_Flat):
    """as fvqa.step.FlatParams: the gradients and, behind them, the error lane in ONE buffer"""
    def __init__(self, rank):
        super().__init__(rank)
        self.grad_store = torch.cat([self.flat_grad, torch.zeros(4)])
        self.flat_grad = self.grad_store[:1000]
        self.err_lane = self.grad_store[1000:1001]


class _ModuleLane(_Module):
    def __init__(self, rank):
        torch.nn.Module.__init__(self)
        self._f = _FlatLane(rank)


def _lane_worker(rank, world, port, out_dir):
    from fvqa.parallel import DataParallel
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    m = _ModuleLane(rank)
    net = DataParallel(m)
    out = {}
    for step, faulty in enumerate((None, 1, None)):                  # step 1: rank 1's persistent GEMM timed out
        word = torch.tensor([1 if faulty == rank else 0], dtype=torch.int64).view(torch.uint8)
        net.error_word = lambda w=word: w
        g_local = m.flat_params().flat_grad.clone()
        net.sync_grads()
        out[step] = dict(lane=float(m.flat_params().err_lane[0]), grad=m.flat_params().flat_grad.clone(), g_local=g_local)
    torch.save(out, os.path.join(out_dir, f"lane{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_gemm_error_word_of_one_rank_reaches_every_rank_with_the_gradients(tmp_path):
    """A timed-out split-K exchange on ONE rank (its workspace's error word) must make ALL replicas skip the step: the flag
    rides as one more element of the gradient all-reduce (FlatParams.err_lane; fvqa_grad_unscale_norm turns a non-zero lane
    into found_inf = 2 on every rank). Checked here: the lane after sync_grads is 0 / non-zero / 0 again on BOTH ranks for a
    clean / faulty-on-rank-1 / clean step, and the gradients next to it are still the plain sum."""
    world, port = 2, _free_port()
    mp.spawn(_lane_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    r0, r1 = torch.load(tmp_path / "lane0.pt"), torch.load(tmp_path / "lane1.pt")
    for step, want in ((0, 0.0), (1, 1.0), (2, 0.0)):
        assert r0[step]["lane"] == want and r1[step]["lane"] == want
        assert torch.equal(r0[step]["grad"], r1[step]["grad"])
        assert torch.allclose(r0[step]["grad"], r0[step]["g_local"] + r1[step]["g_local"])
