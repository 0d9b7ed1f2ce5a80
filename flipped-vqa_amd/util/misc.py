"""Trainer utilities with the reference's names (reference util/misc.py): meters + logger,
distributed init, the loss scaler with gradient-norm count, trainable-only checkpoints.

Only what the training entry points call is provided; eval-side helpers (per-qtype accuracy,
save_result) belong to the validation path, which is outside the MI355X hot path.
"""
from __future__ import annotations

import builtins
import datetime
import os
import time
from collections import defaultdict, deque
from pathlib import Path

import torch
import torch.distributed as dist

from fvqa import ops
from fvqa.synth import is_trainable


# ------------------------------------------------------------------------------ meters
class SmoothedValue:
    """Windowed median/avg plus a global average (reference util/misc.py:27-100)."""

    def __init__(self, window_size=20, fmt=None):
        self.window = deque(maxlen=window_size)
        self.fmt = fmt or "{median:.4f} ({global_avg:.4f})"
        self.total = 0.0
        self.count = 0

    def update(self, value, n=1):
        self.window.append(value)
        self.count += n
        self.total += value * n

    def synchronize_between_processes(self):
        """Sum (count, total) over ranks — the window is not synchronised (util/misc.py:58-70)."""
        if not is_dist_avail_and_initialized():
            return
        dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
        t = torch.tensor([self.count, self.total], dtype=torch.float64, device=dev)
        dist.barrier()
        dist.all_reduce(t)
        self.count, self.total = int(t[0].item()), float(t[1].item())

    @property
    def median(self):
        return torch.tensor(list(self.window)).median().item()

    @property
    def avg(self):
        return torch.tensor(list(self.window), dtype=torch.float32).mean().item()

    @property
    def global_avg(self):
        return self.total / self.count if self.count else 0

    @property
    def max(self):
        return max(self.window)

    @property
    def value(self):
        return self.window[-1]

    def __str__(self):
        if not self.window:
            return "n/a"
        return self.fmt.format(median=self.median, avg=self.avg, global_avg=self.global_avg, max=self.max,
                               value=self.value)


class MetricLogger:
    def __init__(self, delimiter="\t"):
        self.meters = defaultdict(SmoothedValue)
        self.delimiter = delimiter

    def update(self, n=1, **kwargs):
        for k, v in kwargs.items():
            if v is None:
                continue
            if isinstance(v, torch.Tensor):
                v = v.item()
            self.meters[k].update(float(v), n=n)

    def add_meter(self, name, meter):
        self.meters[name] = meter

    def __getattr__(self, attr):
        meters = self.__dict__.get("meters", {})
        if attr in meters:
            return meters[attr]
        raise AttributeError(attr)

    def __str__(self):
        return self.delimiter.join(f"{k}: {m}" for k, m in self.meters.items())

    def synchronize_between_processes(self):
        for m in self.meters.values():
            m.synchronize_between_processes()

    def log_every(self, iterable, print_freq, header=""):
        """Yield items; print progress every `print_freq` iterations (and on the last one).
        print_freq <= 0 is clamped to 1: the reference divides by it (util/misc.py:152) and so
        crashes on loaders shorter than 4 batches."""
        print_freq = max(1, int(print_freq))
        n = len(iterable)
        iter_time, data_time = SmoothedValue(fmt="{avg:.4f}"), SmoothedValue(fmt="{avg:.4f}")
        start = end = time.time()
        for i, obj in enumerate(iterable):
            data_time.update(time.time() - end)
            yield obj
            iter_time.update(time.time() - end)
            if i % print_freq == 0 or i == n - 1:
                eta = datetime.timedelta(seconds=int(iter_time.global_avg * (n - i)))
                msg = [header, f"[{i:>{len(str(n))}}/{n}]", f"eta: {eta}", str(self), f"time: {iter_time}",
                       f"data: {data_time}"]
                if torch.cuda.is_available():
                    msg.append(f"max mem: {torch.cuda.max_memory_allocated() / 2**20:.0f}")
                print(self.delimiter.join(msg))
            end = time.time()
        total = time.time() - start
        print(f"{header} Total time: {datetime.timedelta(seconds=int(total))} ({total / max(n, 1):.4f} s / it)")


# ------------------------------------------------------------------------------ distributed
def is_dist_avail_and_initialized():
    return dist.is_available() and dist.is_initialized()


def get_world_size():
    return dist.get_world_size() if is_dist_avail_and_initialized() else 1


def get_rank():
    return dist.get_rank() if is_dist_avail_and_initialized() else 0


def is_main_process():
    return get_rank() == 0


def save_on_master(*args, **kwargs):
    if is_main_process():
        torch.save(*args, **kwargs)


def setup_for_distributed(is_master):
    """Rank-0-only print with a timestamp; print(..., force=True) prints everywhere."""
    raw = builtins.print

    def gated(*args, **kwargs):
        force = kwargs.pop("force", False) or get_world_size() > 8
        if is_master or force:
            raw(f"[{datetime.datetime.now().time()}] ", end="")
            raw(*args, **kwargs)

    builtins.print = gated


def init_distributed_mode(args):
    """One process per GPU; RANK/WORLD_SIZE/LOCAL_RANK from torchrun (or OpenMPI / SLURM), `nccl`
    backend == RCCL over xGMI on ROCm (reference util/misc.py:220-250)."""
    env = os.environ
    if getattr(args, "dist_on_itp", False):
        args.rank, args.world_size = int(env["OMPI_COMM_WORLD_RANK"]), int(env["OMPI_COMM_WORLD_SIZE"])
        args.gpu = int(env["OMPI_COMM_WORLD_LOCAL_RANK"])
        args.dist_url = f"tcp://{env['MASTER_ADDR']}:{env['MASTER_PORT']}"
        env["LOCAL_RANK"], env["RANK"], env["WORLD_SIZE"] = str(args.gpu), str(args.rank), str(args.world_size)
    elif "RANK" in env and "WORLD_SIZE" in env:
        args.rank, args.world_size, args.gpu = int(env["RANK"]), int(env["WORLD_SIZE"]), int(env["LOCAL_RANK"])
    elif "SLURM_PROCID" in env:
        args.rank = int(env["SLURM_PROCID"])
        args.gpu = args.rank % torch.cuda.device_count()
    else:
        print("Not using distributed mode")
        setup_for_distributed(is_master=True)
        args.distributed = False
        args.gpu = getattr(args, "gpu", 0)
        return
    args.distributed = True
    torch.cuda.set_device(args.gpu)
    args.dist_backend = "nccl"
    print(f"| distributed init (rank {args.rank}): {args.dist_url}, gpu {args.gpu}", flush=True)
    dist.init_process_group(backend=args.dist_backend, init_method=args.dist_url, world_size=args.world_size,
                            rank=args.rank)
    # start-up self-diagnosis (fvqa/rankcheck.py): one rank per whole MI355X, host threads pinned per rank — a wrong
    # LOCAL_RANK -> device map or a partitioned device is reported here, by every rank, before the first step
    from fvqa import rankcheck
    args.host_threads = rankcheck.pin_host_threads(args.world_size, getattr(args, "num_workers", 0))
    diag = rankcheck.check_ranks(args.world_size, args.rank, args.gpu, args.gpu,
                                 rehearsal=os.environ.get("FVQA_DIST_REHEARSAL") == "1")
    setup_for_distributed(args.rank == 0)
    print(f"| {len(diag['reports'])} ranks, one per device: " +
          ", ".join(f"r{r['rank']}@{r['host']}:{r.get('device_id') or r.get('device_index')}({r.get('cu_count')} CUs)"
                    for r in diag["reports"]) + f"; {args.host_threads} host threads per rank", flush=True)


# ------------------------------------------------------------------------------ loss scaler
def get_grad_norm_(parameters, norm_type: float = 2.0) -> torch.Tensor:
    """Norm of the per-parameter gradient norms (reference util/misc.py:282-294)."""
    if isinstance(parameters, torch.Tensor):
        parameters = [parameters]
    grads = [p.grad.detach() for p in parameters if p.grad is not None]
    if not grads:
        return torch.tensor(0.0)
    if norm_type == float("inf"):
        return max(g.abs().max() for g in grads)
    return torch.norm(torch.stack([torch.norm(g, norm_type) for g in grads]), norm_type)


class NativeScalerWithGradNormCount:
    """Dynamic loss scaling + grad-norm + optimizer step, with torch GradScaler's state dict
    (reference util/misc.py:253-279). With fvqa.optim.FusedAdamW the whole boundary step —
    gradient all-reduce, unscale, inf check, norm, AdamW, scale update — runs as device kernels
    with no device->host read; with any other optimizer it falls back to torch.amp.GradScaler."""

    state_dict_key = "amp_scaler"

    def __init__(self, init_scale=65536.0, growth_factor=2.0, backoff_factor=0.5, growth_interval=2000,
                 enabled=True):
        self.growth_factor, self.backoff_factor, self.growth_interval = growth_factor, backoff_factor, growth_interval
        self.enabled = enabled
        self._init_scale = init_scale if enabled else 1.0
        self._init_tracker = 0
        self._dev = None
        self._torch_scaler = None

    def _lazy(self, device):
        if self._dev is None:
            f = dict(dtype=torch.float32, device=device)
            self._scale = torch.full((1,), self._init_scale, **f)
            self._tracker = torch.full((1,), float(self._init_tracker), **f)
            self._found = torch.zeros(1, **f)
            self._norm = torch.zeros(1, **f)
            self._dev = device

    def __call__(self, loss, optimizer, clip_grad=None, parameters=None, create_graph=False, update_grad=True):
        from fvqa.optim import FusedAdamW
        if not isinstance(optimizer, FusedAdamW):
            return self._generic(loss, optimizer, clip_grad, parameters, create_graph, update_grad)
        self._lazy(loss.device)
        (loss * self._scale).sum().backward(create_graph=create_graph)
        if not update_grad:
            return None
        flat = optimizer.flat
        div = 1
        if optimizer.grad_sync is not None:
            r = optimizer.grad_sync()            # all-reduce(SUM) -> replica count; the mean's 1/world rides in the unscale kernel
            div = r if isinstance(r, int) and r > 1 else 1
        n_seg = flat.seg_off.numel() - 1
        if getattr(self, "_ws", None) is None or self._seg_sq.numel() < n_seg:
            self._seg_sq = torch.empty(n_seg, dtype=torch.float32, device=self._dev)
            self._ws = torch.empty(ops.grad_norm_workspace(n_seg), dtype=torch.uint8, device=self._dev)
        # a timed-out split-K exchange (error word of the stream's GEMM workspace) makes found_inf 2: skipped like an overflow
        # ... on THIS rank through the error word, on any rank through the error lane summed in with the gradients
        lane = getattr(flat, "err_lane", None) if optimizer.grad_sync is not None else None
        ops.grad_unscale_norm(flat.flat_grad, flat.seg_off, self._scale, self._seg_sq, self._found, self._norm,
                              self._ws, grad_div=float(div), gemm_err=ops.gemm_error_word(self._dev), err_lane=lane)
        if clip_grad is not None:
            flat.flat_grad.mul_(torch.clamp(clip_grad / (self._norm + 1e-6), max=1.0))
        optimizer.step(found_inf=self._found)
        scale = self._scale if self.enabled else None
        ops.scaler_update(optimizer.step_dev, scale, self._tracker if self.enabled else None, self._found,
                          self.growth_factor, self.backoff_factor, self.growth_interval)
        return self._norm[0]

    def _generic(self, loss, optimizer, clip_grad, parameters, create_graph, update_grad):
        if self._torch_scaler is None:
            self._torch_scaler = torch.amp.GradScaler("cuda", init_scale=self._init_scale, enabled=self.enabled)
        s = self._torch_scaler
        s.scale(loss).sum().backward(create_graph=create_graph)
        if not update_grad:
            return None
        s.unscale_(optimizer)
        if clip_grad is not None:
            norm = torch.nn.utils.clip_grad_norm_(parameters, clip_grad)
        else:
            norm = get_grad_norm_(parameters)
        s.step(optimizer)
        s.update()
        return norm

    def state_dict(self):
        if self._torch_scaler is not None:
            return self._torch_scaler.state_dict()
        scale = float(self._scale.item()) if self._dev is not None else self._init_scale
        tracker = int(self._tracker.item()) if self._dev is not None else self._init_tracker
        return {"scale": scale, "growth_factor": self.growth_factor, "backoff_factor": self.backoff_factor,
                "growth_interval": self.growth_interval, "_growth_tracker": tracker}

    def load_state_dict(self, state_dict):
        if not state_dict:
            return
        self._init_scale = float(state_dict["scale"])
        self.growth_factor = state_dict.get("growth_factor", self.growth_factor)
        self.backoff_factor = state_dict.get("backoff_factor", self.backoff_factor)
        self.growth_interval = state_dict.get("growth_interval", self.growth_interval)
        self._init_tracker = int(state_dict.get("_growth_tracker", 0))    # applied now or at the first call (lazy device state)
        if self._dev is not None:
            self._scale.fill_(self._init_scale)
            self._tracker.fill_(float(self._init_tracker))


# ------------------------------------------------------------------------------ checkpoints
def trainable_state(model_without_ddp):
    """name -> tensor for the parameters the freeze policy trains (reference llama_vqa.py:72,
    util/misc.py:303-306) — detached copies, so a checkpoint never aliases the flat buffer."""
    return {n: p.detach().clone() for n, p in model_without_ddp.named_parameters() if is_trainable(n)}


def save_model(args, epoch, model, model_without_ddp, optimizer, loss_scaler, name):
    """`<output_dir>/<name>.pth` with the reference's layout: model (trainables only), optimizer,
    epoch, scaler, args (reference util/misc.py:297-320)."""
    path = Path(args.output_dir) / f"{name}.pth"
    payload = {"model": trainable_state(model_without_ddp), "optimizer": optimizer.state_dict(), "epoch": epoch,
               "scaler": loss_scaler.state_dict() if loss_scaler is not None else None, "args": args}
    save_on_master(payload, path)


def load_model(args, model_without_ddp, optimizer, loss_scaler):
    """--resume: trainables (strict=False), then optimizer + scaler + start_epoch
    (reference util/misc.py:322-336). URL checkpoints need network access and are refused."""
    if not getattr(args, "resume", ""):
        return
    if str(args.resume).startswith("https"):
        raise RuntimeError("URL checkpoints cannot be fetched here; pass a local path to --resume")
    ckpt = torch.load(args.resume, map_location="cpu", weights_only=False)
    own = dict(model_without_ddp.named_parameters())
    with torch.no_grad():
        for n, t in ckpt["model"].items():
            if n in own:
                own[n].copy_(t.to(own[n].device, own[n].dtype))
    print(f"Resume checkpoint {args.resume}")
    if "optimizer" in ckpt and "epoch" in ckpt and not getattr(args, "eval", False):
        optimizer.load_state_dict(ckpt["optimizer"])
        args.start_epoch = ckpt["epoch"] + 1
        if ckpt.get("scaler") is not None:
            loss_scaler.load_state_dict(ckpt["scaler"])
        print("With optim & sched!")


def all_reduce_mean(value: float) -> float:
    if get_world_size() == 1:
        return value
    t = torch.tensor(value, device="cuda" if dist.get_backend() == "nccl" else "cpu")
    dist.all_reduce(t)
    return (t / get_world_size()).item()


# ---- validation bookkeeping (reference util/misc.py:343-600) -----------------------------------------
NEXTQA_GROUPS = {"C": (1, 2), "T": (3, 4, 5), "D": (6, 7, 8)}       # causal / temporal / descriptive type ids


def log_qtype(data, hit, metric_logger: MetricLogger, args):
    """Per-question-type accuracy meters for NExT-QA: C (CH, CW), T (TN, TC, TP), D (DL, DC, DO), Total — each
    updated with weight = number of questions of the group in this batch (reference util/misc.py:443-449,526-532).
    Other datasets of the reference are not built."""
    if getattr(args, "dataset", "nextqa") != "nextqa":
        return
    eps = 1e-10
    qtype = torch.as_tensor(data["qtype"]).cpu()
    hit = torch.as_tensor(hit).cpu().to(torch.float64)
    for name, ids in NEXTQA_GROUPS.items():
        sel = torch.zeros_like(qtype, dtype=torch.bool)
        for i in ids:
            sel |= qtype == i
        n = float(sel.sum())
        metric_logger.update(n=n + eps, **{name: float(hit[sel].sum()) / (n + eps)})
    n = float(qtype.numel())
    metric_logger.update(n=n + eps, Total=float(hit.sum()) / n if n else 0.0)


def save_result(result, result_dir, filename):
    """Each rank writes `<filename>_rank<r>.json`; rank 0 concatenates them into `<filename>.json`
    (reference util/misc.py:570-600, JSON list form)."""
    import json
    mine = os.path.join(result_dir, "%s_rank%d.json" % (filename, get_rank()))
    with open(mine, "w") as f:
        json.dump(result, f, default=lambda o: o.tolist() if hasattr(o, "tolist") else str(o))
    if is_dist_avail_and_initialized():
        dist.barrier()
    if is_main_process():
        merged = []
        for r in range(get_world_size()):
            with open(os.path.join(result_dir, "%s_rank%d.json" % (filename, r))) as f:
                merged += json.load(f)
        with open(os.path.join(result_dir, "%s.json" % filename), "w") as f:
            json.dump(merged, f)
    return os.path.join(result_dir, "%s.json" % filename)
