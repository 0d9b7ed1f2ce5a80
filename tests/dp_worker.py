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

N_STEPS, INF_STEP, INF_RANK = 4, 1, 1


def batch_seed(rank, world, i):
    return 100 + rank + world * i


def perturb_trainables(model, seed):
    """What per-rank seeding does to the randomly initialised trainables (train.py:87 seeds seed+rank)."""
    g = torch.Generator(device="cpu").manual_seed(seed)
    flat = model.flat_params().flat
    flat.add_(0.01 * torch.randn(flat.numel(), generator=g).to(flat.device))


def main():
    rank, world, port, out_dir = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    import util.misc as misc
    from fvqa import synth
    from fvqa.optim import FusedAdamW, param_groups_weight_decay
    from fvqa.parallel import DataParallel
    from tests.gpu_util import build_model

    cfg = synth.preset("tiny", vaq=True, qav=True)
    model, args = build_model(cfg, torch.float32)
    perturb_trainables(model, seed=1000 + rank)                 # replicas start DIFFERENT ...
    opt = FusedAdamW(param_groups_weight_decay(model, args.weight_decay), lr=0.01, betas=(0.9, 0.95),
                     flat=model.flat_params())
    net = DataParallel(model)                                   # ... and must leave here identical
    flat = model.flat_params()
    trace = {"p_init": flat.flat.detach().cpu().clone()}
    scaler = misc.NativeScalerWithGradNormCount()
    step_no = {"i": 0}

    def sync():
        if step_no["i"] == INF_STEP and rank == INF_RANK:       # an overflow on ONE rank
            flat.flat_grad[7] = float("inf")
        net.sync_grads()

    opt.grad_sync = sync
    for i in range(N_STEPS):
        step_no["i"] = i
        opt.zero_grad()
        a, b, c = net(synth.make_batch(cfg, seed=batch_seed(rank, world, i)))
        scaler(a + b + c, opt, parameters=None, update_grad=True)
        torch.cuda.synchronize()
        trace[f"p{i}"] = flat.flat.detach().cpu().clone()
        trace[f"scale{i}"] = float(scaler._scale.item())
        trace[f"found{i}"] = float(scaler._found.item())
        trace[f"step{i}"] = float(opt.step_dev.item())
    torch.save(trace, os.path.join(out_dir, f"rank{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
