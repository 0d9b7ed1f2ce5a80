"""Per-iteration learning-rate rule of the training loop: linear warm-up over `warmup_epochs`,
then a half cosine from `lr` down to `min_lr` at `epochs` (same rule and call signature as
reference util/lr_sched.py:9-21; `epoch` is fractional: step/len(loader) + epoch)."""
import math


def lr_at(epoch: float, lr: float, min_lr: float, warmup_epochs: float, epochs: float) -> float:
    if epoch < warmup_epochs:
        return lr * epoch / warmup_epochs
    progress = (epoch - warmup_epochs) / (epochs - warmup_epochs)
    return min_lr + 0.5 * (lr - min_lr) * (1.0 + math.cos(math.pi * progress))


def adjust_learning_rate(optimizer, epoch, args):
    value = lr_at(epoch, args.lr, args.min_lr, args.warmup_epochs, args.epochs)
    for group in optimizer.param_groups:
        group["lr"] = value * group.get("lr_scale", 1.0)
    return value
