"""One rank of the 2-rank data-parallel rehearsal (tests/test_dp_gpu.py): train.py's wiring —
DataParallel (broadcast + sync_grads) + NativeScalerWithGradNormCount + FusedAdamW on the HIP step —
with both ranks on ONE GPU over gloo. Run as: python tests/dp_worker.py <rank> <world> <port> <out_dir>."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "flipped-vqa_amd"), ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)
os.environ.setdefault("FVQA_SYNTHETIC_TOKENIZER", "1")

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

N_STEPS, INF_STEP, INF_RANK = 5, 1, 1
ERR_STEP, ERR_RANK = 3, 0          # a timed-out split-K exchange (persistent-GEMM error word) on ONE rank
SIDE_STEP = 2                      # forward + backward on a side stream, optimizer step on the default one


def scaler_scale(scaler, loss):
    scaler._lazy(loss.device)
    return scaler._scale


def finish_step(scaler, opt):
    """What NativeScalerWithGradNormCount.__call__ does after its backward (util/misc.py; reference util/misc.py:259-273):
    gradient sync, unscale + norm, AdamW, scale update — on the CURRENT stream."""
    from fvqa import ops
    flat = opt.flat
    r = opt.grad_sync()
    div = r if isinstance(r, int) and r > 1 else 1
    n_seg = flat.seg_off.numel() - 1
    if getattr(scaler, "_ws", None) is None or scaler._seg_sq.numel() < n_seg:
        scaler._seg_sq = torch.empty(n_seg, dtype=torch.float32, device=scaler._dev)
        scaler._ws = torch.empty(ops.grad_norm_workspace(n_seg), dtype=torch.uint8, device=scaler._dev)
    ops.grad_unscale_norm(flat.flat_grad, flat.seg_off, scaler._scale, scaler._seg_sq, scaler._found, scaler._norm, scaler._ws,
                          grad_div=float(div), gemm_err=ops.gemm_error_word(scaler._dev), err_lane=getattr(flat, "err_lane", None))
    opt.step(found_inf=scaler._found)
    ops.scaler_update(opt.step_dev, scaler._scale, scaler._tracker, scaler._found, scaler.growth_factor, scaler.backoff_factor,
                      scaler.growth_interval)


def batch_seed(rank, world, i):
    return 100 + rank + world * i


def perturb_trainables(model, seed):
    """What per-rank seeding does to the randomly initialised trainables (train.py:87 seeds seed+rank)."""
    g = torch.Generator(device="cpu").manual_seed(seed)
    flat = model.flat_params().flat
    flat.add_(0.01 * torch.randn(flat.numel(), generator=g).to(flat.device))


def rccl_one_rank(port, out_dir):
    """RCCL beside libfvqa_hip.so on one GPU: a ONE-rank `nccl` process group (nccl == RCCL on ROCm), train.py's wiring
    (DataParallel broadcast + all-reduce of the real flat gradient buffer of a 7B-width model, forced although world
    is 1, + loss scaler + FusedAdamW) for 3 steps on the HIP step — bitwise equal to the same steps without any process
    group. Reference: train.py:104-117, util/misc.py:220-250."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=port, FVQA_DP_FORCE_ALLREDUCE="1")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(0)
    import util.misc as misc
    from fvqa import synth
    from fvqa.optim import FusedAdamW, param_groups_weight_decay
    from fvqa.parallel import DataParallel
    from tests.gpu_util import build_model

    cfg = synth.preset("7b_l2", batch_size=2)

    def run(dp):
        model, args = build_model(cfg, torch.bfloat16)
        flat = model.flat_params()
        opt = FusedAdamW(param_groups_weight_decay(model, args.weight_decay), lr=0.01, betas=(0.9, 0.95), flat=flat)
        net = model
        if dp:
            net = DataParallel(model)                           # broadcast over RCCL
            net.comm_events = []
            opt.grad_sync = net.sync_grads
        scaler = misc.NativeScalerWithGradNormCount()
        out = []
        for i in range(3):
            opt.zero_grad()
            a, b, c = net(synth.make_batch(cfg, seed=200 + i))
            scaler(a + b + c, opt, parameters=None, update_grad=True)
            torch.cuda.synchronize()
            out.append(flat.flat.detach().cpu().clone())
        model._engine.check_gemm_error()
        ms = [e0.elapsed_time(e1) for e0, e1 in net.comm_events] if dp else []
        return out, flat.flat_grad.numel() * 4, ms

    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    backend, ranks = dist.get_backend(), dist.get_world_size()
    with_dp, nbytes, ms = run(True)
    dist.barrier()
    dist.destroy_process_group()
    plain, _, _ = run(False)
    torch.save(dict(with_dp=with_dp, plain=plain, backend=backend, ranks=ranks, allreduce_bytes=nbytes, allreduce_ms=ms),
               os.path.join(out_dir, "rccl1.pt"))


def main():
    if len(sys.argv) > 5 and sys.argv[5] == "rccl1":
        return rccl_one_rank(sys.argv[3], sys.argv[4])
    rank, world, port, out_dir = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    # a box with a device per rank runs the real thing — one rank per GPU, `nccl` (= RCCL) —, a one-GPU box the rehearsal:
    # both ranks on device 0 over gloo (the tiny fp32 model never launches the persistent split-K kernel, which needs a
    # device to itself, so the two processes may share it)
    multi = torch.cuda.device_count() >= world and os.environ.get("FVQA_DP_TEST_BACKEND", "auto") != "gloo"
    device_index = rank if multi else 0
    torch.cuda.set_device(device_index)
    if multi:
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", device_index))
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    from fvqa import rankcheck
    diag = rankcheck.check_ranks(world, rank, rank, device_index, rehearsal=not multi, strict=True)
    import util.misc as misc
    from fvqa import synth
    from fvqa.optim import FusedAdamW, param_groups_weight_decay
    from fvqa.parallel import DataParallel
    from tests.gpu_util import build_model

    cfg = synth.preset("tiny", vaq=True, qav=True)
    model, args = build_model(cfg, torch.float32, device=f"cuda:{device_index}")
    perturb_trainables(model, seed=1000 + rank)                 # replicas start DIFFERENT ...
    opt = FusedAdamW(param_groups_weight_decay(model, args.weight_decay), lr=0.01, betas=(0.9, 0.95),
                     flat=model.flat_params())
    net = DataParallel(model)                                   # ... and must leave here identical
    flat = model.flat_params()
    trace = {"p_init": flat.flat.detach().cpu().clone(), "backend": dist.get_backend(), "multi_device": multi,
             "problems": diag["problems"]}
    scaler = misc.NativeScalerWithGradNormCount()
    step_no = {"i": 0}

    from fvqa import ops
    word = ops.gemm_workspace(torch.device("cuda", device_index))[:8]      # this rank's persistent-GEMM workspace: its error word

    def sync():
        if step_no["i"] == INF_STEP and rank == INF_RANK:       # an overflow on ONE rank
            flat.flat_grad[7] = float("inf")
        word.view(torch.int64)[0] = 1 if (step_no["i"] == ERR_STEP and rank == ERR_RANK) else 0
        return net.sync_grads()

    opt.grad_sync = sync
    # SIDE_STEP: forward + backward run on a SIDE stream, the gradient all-reduce + unscale + AdamW on the default stream with no
    # host synchronisation in between — correct only because sync_grads makes the current stream wait for the stream the
    # gradients were produced on (fvqa/parallel.py); every other step runs everything on one stream
    side = torch.cuda.Stream(device=device_index)
    trace["ordering"] = []
    for i in range(N_STEPS):
        step_no["i"] = i
        opt.zero_grad()
        batch = synth.make_batch(cfg, seed=batch_seed(rank, world, i))
        if i == SIDE_STEP:
            side.wait_stream(torch.cuda.current_stream(device_index))
            with torch.cuda.stream(side):
                a, b, c = net(batch)
                loss = a + b + c
                (loss * scaler_scale(scaler, loss)).sum().backward()
            # (the loss scaler's own backward call is replaced by the one above: the step below only syncs, unscales and steps)
            finish_step(scaler, opt)
        else:
            a, b, c = net(batch)
            scaler(a + b + c, opt, parameters=None, update_grad=True)
        trace["ordering"].append(net.last_sync_ordering)
        torch.cuda.synchronize()
        trace[f"p{i}"] = flat.flat.detach().cpu().clone()
        trace[f"scale{i}"] = float(scaler._scale.item())
        trace[f"found{i}"] = float(scaler._found.item())
        trace[f"step{i}"] = float(opt.step_dev.item())
    word.view(torch.int64)[0] = 0
    model._engine.check_gemm_error()                            # (two persistent grids share this GPU: no exchange timed out)
    torch.save(trace, os.path.join(out_dir, f"rank{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
