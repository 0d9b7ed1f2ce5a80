"""`engine.train_one_epoch` with the reference's signature and return value (reference
engine.py:10-56): per-iteration LR schedule, forward, sum of the three flipped losses, finite
check, gradient accumulation through the loss scaler, meters.

Host-side differences that do not change results: the three loss values come back in ONE
device->host transfer per iteration instead of four `.item()` calls + a full device sync; that
transfer is WAITED FOR only after the backward (and, at an accumulation boundary, the optimizer
step) of the iteration has been launched, so the device never idles while the host reads the
loss (reading first, as the reference does, leaves the GPU idle from the end of the forward until
the host has issued the first backward kernel: 0.5-1 ms of a 29 ms step). A non-finite loss still
prints the reference's message and exits with status 1 before anything else happens on the host;
the one difference is that the backward / optimizer step of THAT iteration has already been queued
on the device when the process exits (a non-finite gradient is skipped by the fused optimizer; a
finite one — e.g. the other two losses when only one CE had no scored token — is applied), which
nothing observes: the reference's train.py does not catch the exit and saves nothing after it. The logging period is clamped to >= 1 (the reference crashes on loaders shorter
than 4 batches).
"""
import math
import sys
from typing import Iterable

import torch

import util.lr_sched as lr_sched
import util.misc as misc


def train_one_epoch(model: torch.nn.Module, data_loader: Iterable, optimizer: torch.optim.Optimizer, epoch: int,
                    loss_scaler, args=None):
    model.train(True)
    log = misc.MetricLogger(delimiter="  ")
    log.add_meter("lr", misc.SmoothedValue(window_size=1, fmt="{value:.6f}"))
    n_iter = len(data_loader)
    accum = args.accum_iter
    optimizer.zero_grad()
    host_vals, copied = None, None

    for it, data in enumerate(log.log_every(data_loader, n_iter // 4, f"Epoch: [{epoch}]")):
        boundary_start = it % accum == 0
        boundary_end = (it + 1) % accum == 0
        if boundary_start:
            lr_sched.adjust_learning_rate(optimizer, it / n_iter + epoch, args)

        vqa_loss, vaq_loss, qav_loss = model(data)
        loss = vqa_loss + vaq_loss + qav_loss
        # the three losses and — riding in the same transfer, at no cost of its own — the found-inf word the loss scaler
        # left on the device in the PREVIOUS iteration (2 = the GEMM error word was set: include/fvqa.h)
        found_dev = getattr(loss_scaler, "_found", None)
        prev_found = found_dev.reshape(()).float() if torch.is_tensor(found_dev) and found_dev.device == vqa_loss.device \
            else torch.zeros((), dtype=torch.float32, device=vqa_loss.device)
        vals_dev = torch.stack([vqa_loss.reshape(()).float(), vaq_loss.reshape(()).float(),
                                qav_loss.reshape(()).float(), prev_found])
        if vals_dev.is_cuda:                                  # one asynchronous D2H copy into pinned memory
            if host_vals is None:
                host_vals, copied = torch.empty(4, dtype=torch.float32, pin_memory=True), torch.cuda.Event()
            host_vals.copy_(vals_dev, non_blocking=True)
            copied.record()

        loss_scaler(loss / accum, optimizer, parameters=model.parameters(), update_grad=boundary_end)

        if vals_dev.is_cuda:
            copied.synchronize()                              # the forward has finished; the backward is queued
            vals = host_vals.tolist()
            if vals[3] == 2.0:                                # (sticky: the error word stays set until someone clears it)
                raise RuntimeError("fvqa: a split-K exchange of the persistent GEMM timed out (found_inf = 2): the "
                                   "previous optimizer step was skipped and the results of its launches are invalid")
        else:
            vals = vals_dev.tolist()
        loss_value = vals[0] + vals[1] + vals[2]
        if not math.isfinite(loss_value):
            print("Loss is {}, stopping training".format(loss_value))
            sys.exit(1)
        if boundary_end:
            optimizer.zero_grad()

        log.update(loss=loss_value, vqa_loss=vals[0], vaq_loss=vals[1], qav_loss=vals[2])
        log.update(lr=optimizer.param_groups[0]["lr"])
        if getattr(args, "debug", False):
            break

    # The LAST iteration's error state. found_inf = 2 is the all-reduced error lane (fvqa/parallel.py: every rank sees it when
    # ANY rank's split-K exchange timed out), so every rank stops here together; reading only the local error word would raise
    # on the faulty rank alone and leave the healthy ones waiting in the meter all-reduce below (round-4 advisor finding).
    found_dev = getattr(loss_scaler, "_found", None)
    if torch.is_tensor(found_dev) and float(found_dev.reshape(-1)[0].item()) == 2.0:
        raise RuntimeError("fvqa: a split-K exchange of the persistent GEMM timed out on some rank (found_inf = 2): the last "
                           "optimizer step was skipped on every rank and the results of its launches are invalid")
    eng = getattr(getattr(model, "module", model), "_engine", None)
    if eng is not None:
        eng.check_gemm_error()                                # this rank's own word (include/fvqa.h: error word)
    log.synchronize_between_processes()
    print("Averaged stats:", log)
    return {k: m.global_avg for k, m in log.meters.items()}


def val_one_epoch(model: torch.nn.Module, data_loader: Iterable, optimizer: torch.optim.Optimizer, epoch: int,
                  args=None):
    """Validation with the reference's signature and return value (reference engine.py:59-145), generation
    mode (`--is_generation_task`, the mode every run script of the reference uses): greedy answer generation,
    nearest-choice matching, accuracy overall and per question type. The per-choice-loss mode calls a model
    branch the reference no longer has (its `inference` returns the generation tuple), so it is rejected."""
    if not getattr(args, "is_generation_task", False):
        raise NotImplementedError("validation is built for --is_generation_task (reference engine.py:78-86)")
    model.eval()
    log = misc.MetricLogger(delimiter="  ")
    log.add_meter("lr", misc.SmoothedValue(window_size=1, fmt="{value:.6f}"))
    for data in log.log_every(data_loader, max(1, len(data_loader) // 4), f"Epoch: [{epoch}]"):
        answer = data["answer"]
        bsz = answer.shape[0]
        with torch.no_grad():
            best, extracted = model(data, inference=True)
        if getattr(args, "output_dir", None):
            import os
            os.makedirs(os.path.join(args.output_dir, "extracted_answers"), exist_ok=True)
            misc.save_result(extracted, os.path.join(args.output_dir, "extracted_answers"),
                             "extracted_answers_epoch%d" % epoch)
        hit = answer.to(best.device) == best
        misc.log_qtype(data, hit, log, args)
        log.update(lr=optimizer.param_groups[0]["lr"])
        log.update(n=bsz, acc=hit.sum().item() / bsz)
        if getattr(args, "debug", False):
            break
    log.synchronize_between_processes()
    print("Averaged stats:", log)
    return {k: m.global_avg for k, m in log.meters.items()}
