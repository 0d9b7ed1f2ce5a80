"""Training entry point with the reference's CLI (reference train.py:24-176): same flags,
`get_args_parser()`, `main(args)`, `validate_args(args)`; launch with
`torchrun --nproc_per_node N train.py ...` (one process per MI355X, RCCL over xGMI).

Extra flags (all optional): --dtype {bf16,fp32}, --random_init, --synthetic, --synthetic_batches N
select closed-form weights / batches so the path runs without LLaMA assets or datasets.
"""
import argparse
import datetime
import json
import os
import time
from pathlib import Path

import numpy as np
import torch

import util.misc as misc
from engine import train_one_epoch
from fvqa import synth
from fvqa.batch_producer import DeviceBatchProducer
from fvqa.optim import FusedAdamW, param_groups_weight_decay
from fvqa.parallel import DataParallel
from llama_vqa import LLaMA_VQA
from util.misc import NativeScalerWithGradNormCount as NativeScaler

# (flag, kwargs) — the reference's flag set, names and defaults (train.py:26-73)
_FLAGS = [
    ("--batch_size", dict(default=64, type=int, help="batch size per GPU (effective = batch_size * accum_iter * #gpus)")),
    ("--epochs", dict(default=400, type=int)),
    ("--accum_iter", dict(default=1, type=int, help="gradient accumulation iterations")),
    ("--llama_model_path", dict(default="./pretrained/llama/", type=str)),
    ("--model", dict(default="llama7B_adapter", type=str, metavar="MODEL")),
    ("--adapter_layer", dict(type=int, default=32, metavar="LENGTH")),
    ("--adapter_len", dict(type=int, default=10, metavar="LENGTH")),
    ("--max_seq_len", dict(type=int, default=512, metavar="LENGTH")),
    ("--max_feats", dict(type=int, default=10, metavar="LENGTH")),
    ("--weight_decay", dict(type=float, default=0.05)),
    ("--lr", dict(type=float, default=None, metavar="LR", help="absolute learning rate")),
    ("--blr", dict(type=float, default=1e-3, metavar="LR", help="base lr: lr = blr * total_batch / 256")),
    ("--min_lr", dict(type=float, default=0.0, metavar="LR")),
    ("--warmup_epochs", dict(type=int, default=40, metavar="N")),
    ("--dataset", dict(default="nextqa", type=str)),
    ("--output_dir", dict(default="./output_dir")),
    ("--device", dict(default="cuda")),
    ("--seed", dict(default=0, type=int)),
    ("--resume", dict(default="")),
    ("--start_epoch", dict(default=0, type=int, metavar="N")),
    ("--num_workers", dict(default=2, type=int)),
    ("--pin_mem", dict(action="store_true")),
    ("--no_pin_mem", dict(action="store_false", dest="pin_mem")),
    ("--world_size", dict(default=1, type=int)),
    ("--local_rank", dict(default=-1, type=int)),
    ("--dist_on_itp", dict(action="store_true")),
    ("--dist_url", dict(default="env://")),
    ("--vaq", dict(action="store_true", help="vaq loss")),
    ("--qav", dict(action="store_true", help="qav loss")),
    ("--bias", dict(type=float, default=3.0, help="attention bias")),
    ("--tau", dict(type=float, default=100.0)),
    ("--sub", dict(action="store_true")),
    ("--is_generation_task", dict(action="store_true")),
    ("--debug", dict(action="store_true")),
    ("--jobid", dict(type=int)),
    ("--audio", dict(action="store_true")),
    ("--audio_only", dict(action="store_true")),
    ("--audio_merge", dict(type=str, choices=["sum", "concat", "attention", "none"], default="none")),
    # MI355X build additions
    ("--dtype", dict(type=str, choices=["bf16", "fp32"], default="bf16", help="storage dtype of frozen weights/activations")),
    ("--random_init", dict(action="store_true", help="closed-form weights instead of a checkpoint")),
    ("--synthetic", dict(action="store_true", help="synthetic tokenizer constants + synthetic batches")),
    ("--synthetic_batches", dict(type=int, default=8, help="batches per epoch with --synthetic")),
]


def get_args_parser():
    parser = argparse.ArgumentParser("Flipped-VQA training (MI355X)", add_help=False)
    for flag, kw in _FLAGS:
        parser.add_argument(flag, **kw)
    parser.set_defaults(pin_mem=True)
    return parser


def validate_args(args):
    """Audio flag consistency (reference train.py:154-168)."""
    assert isinstance(args.audio, bool) and isinstance(args.audio_only, bool)
    if args.audio_only:
        assert args.audio, "If audio_only is True, audio must also be set to True"
    if args.audio and args.audio_only:
        assert args.audio_merge == "none", "If you only need audio, you should not specify merge method"
        args.audio_merge = None
    if args.audio and not args.audio_only:
        assert args.audio_merge in ("sum", "concat", "attention"), \
            "An audio_merge method must be specified if audio is True and audio_only is False"


def build_loaders(args, model):
    if args.synthetic or args.dataset == "synthetic":
        p = model.params
        cfg = synth.SynthConfig(dim=p.dim, n_heads=p.n_heads, n_layers=p.n_layers, vocab_size=model.vocab_size,
                                max_feats=args.max_feats, max_seq_len=args.max_seq_len, batch_size=args.batch_size,
                                vaq=args.vaq, qav=args.qav)
        return synth.SyntheticLoader(cfg, args.synthetic_batches, misc.get_rank(), misc.get_world_size(), pin=True), None
    from dataloader import load_data          # NExT-QA reader + collate of this package (dataloader/)
    return load_data(args, model.tokenizer, split="train"), load_data(args, model.tokenizer, split="val")


def main(args):
    misc.init_distributed_mode(args)
    print("job dir: {}".format(os.path.dirname(os.path.realpath(__file__))))
    print("{}".format(args).replace(", ", ",\n"))
    torch.cuda.set_device(args.gpu)

    seed = args.seed + misc.get_rank()
    torch.manual_seed(seed)
    np.random.seed(seed)

    model = LLaMA_VQA(args)
    model.to(torch.device("cuda", args.gpu))
    model_without_ddp = model
    train_loader, val_loader = build_loaders(args, model)

    eff_batch_size = args.batch_size * args.accum_iter * misc.get_world_size()
    if args.lr is None:
        args.lr = args.blr * eff_batch_size / 256
    print("base lr: %.2e" % (args.lr * 256 / eff_batch_size))
    print("actual lr: %.2e" % args.lr)
    print("accumulate grad iterations: %d" % args.accum_iter)
    print("effective batch size: %d" % eff_batch_size)

    optimizer = FusedAdamW(param_groups_weight_decay(model_without_ddp, args.weight_decay), lr=args.lr,
                           betas=(0.9, 0.95), flat=model_without_ddp.flat_params())
    if args.distributed:
        model = DataParallel(model)
        optimizer.grad_sync = model.sync_grads
    print(optimizer)
    loss_scaler = NativeScaler()
    misc.load_model(args=args, model_without_ddp=model_without_ddp, optimizer=optimizer, loss_scaler=loss_scaler)
    if args.distributed:                       # whatever each rank loaded, replicas start from rank 0's state
        model.broadcast_params()
        model.broadcast_optimizer(optimizer, loss_scaler)

    print(f"Start training for {args.epochs} epochs")
    t0 = time.time()
    best_acc = 0.0
    for epoch in range(args.start_epoch, args.epochs):
        sampler = getattr(train_loader, "sampler", None)
        if args.distributed and hasattr(sampler, "set_epoch"):
            sampler.set_epoch(epoch)
        # pinned single-copy H2D staging on a side stream: the step never waits for a pageable copy
        staged = DeviceBatchProducer(train_loader, torch.device("cuda", args.gpu), depth=3)
        train_stats = train_one_epoch(model, staged, optimizer, epoch, loss_scaler, args=args)
        log_stats = {**{f"train_{k}": v for k, v in train_stats.items()}, "epoch": epoch}
        val_stats = None
        if val_loader is not None:
            try:
                from engine import val_one_epoch        # provided by the reference checkout, if present
                val_stats = val_one_epoch(model_without_ddp, val_loader, optimizer, epoch, args=args)
            except ImportError:
                val_stats = None
        if val_stats is not None:
            log_stats.update({f"val_{k}": v for k, v in val_stats.items()})
        improved = val_stats is None or best_acc < val_stats["acc"]
        if args.output_dir and improved:
            best_acc = val_stats["acc"] if val_stats is not None else best_acc
            misc.save_model(args=args, model=model, model_without_ddp=model_without_ddp, optimizer=optimizer,
                            loss_scaler=loss_scaler, epoch=epoch, name="checkpoint_best")
        if args.output_dir and misc.is_main_process():
            with open(os.path.join(args.output_dir, "log.txt"), mode="a", encoding="utf-8") as f:
                f.write(json.dumps(log_stats) + "\n")
    print("Training time {}".format(str(datetime.timedelta(seconds=int(time.time() - t0)))))


if __name__ == "__main__":
    args = get_args_parser().parse_args()
    validate_args(args)
    if args.output_dir:
        Path(args.output_dir).mkdir(parents=True, exist_ok=True)
    main(args)
