"""Helpers for the -m gpu parity tests: build the product model from a closed-form config and
pull logits / layer outputs / gradients out of the step engine's arena."""
import types

import torch

import llama
from fvqa import synth
from llama_vqa import fill_closed_form


def make_args(cfg, **over):
    a = types.SimpleNamespace(
        max_feats=cfg.max_feats, bias=cfg.bias, tau=cfg.tau, llama_model_path="/nonexistent/", vaq=cfg.vaq,
        qav=cfg.qav, synthetic=True, vocab_size=cfg.vocab_size, audio=False, audio_only=False, audio_merge="none",
        debug=False, adapter_len=cfg.adapter_len, adapter_layer=cfg.adapter_layer, max_seq_len=cfg.max_seq_len,
        accum_iter=1, lr=1e-3, min_lr=0.0, warmup_epochs=1, epochs=4, weight_decay=0.1)
    for k, v in over.items():
        setattr(a, k, v)
    return a


def build_model(cfg, dtype=torch.float32, device="cuda"):
    args = make_args(cfg)
    ma = llama.ModelArgs(max_seq_len=cfg.max_seq_len, max_batch_size=32, adapter_len=cfg.adapter_len,
                         adapter_layer=cfg.adapter_layer, **cfg.params_json())
    ma.vocab_size = cfg.vocab_size
    prev = torch.get_default_dtype()
    torch.set_default_dtype(dtype)
    try:
        with torch.device(device):
            model = llama.Transformer(ma, args)
    finally:
        torch.set_default_dtype(prev)
    for n, p in model.named_parameters():
        p.requires_grad = synth.is_trainable(n)
        if p.requires_grad:
            p.data = p.data.float()
    fill_closed_form(model, cfg)           # after the fp32 cast: trainables keep full-precision values
    return model, args


def run_step(model, batch, loss_weights=(1.0, 1.0, 1.0), lm_head="all"):
    """forward + backward of w·losses; returns losses dict, grads dict (CPU fp32), logits dict, layer outs.

    lm_head: "all" = the reference's dense form — every projection of the last layer and the LM head at every position (the parity
    tests compare the logits of every row); "scored" = the product default: the last layer's post-attention half, the heads and their
    backward on the rows a head reads only (fvqa/step.py TailRows) — the logits and the last layer's output then hold NaN rows
    everywhere else."""
    model.ensure_engine().lm_head_rows = lm_head
    flat = model.flat_params()
    flat.zero_grad()
    vqa, vaq, qav = model(batch)
    total = vqa * loss_weights[0]
    if model.args.vaq:
        total = total + vaq * loss_weights[1]
    if model.args.qav:
        total = total + qav * loss_weights[2]
    total.sum().backward()
    torch.cuda.synchronize()
    eng = model._engine
    B = batch["video"].shape[0]
    S = batch["text_id"]["vqa"].shape[-1]
    ar = eng.arena(eng.n_streams * B, S)
    losses = {"vqa": float(vqa.detach()), "vaq": float(vaq.detach()) if model.args.vaq else 0.0,
              "qav": float(qav.detach()) if model.args.qav else 0.0}
    grads = {n: p.grad.detach().float().cpu().clone() for n, p in model.named_parameters() if p.requires_grad}
    logits = {}
    sc = eng.last_scored
    for k, t in enumerate(eng.tasks):
        if t == "qav":
            continue
        if sc is None:
            logits[t] = ar.logits[k * B * S:(k + 1) * B * S].view(B, S, -1).float().cpu()
        else:
            o0, m = sc.segs[k][0], sc.counts[k]
            full = torch.full((B * S, eng.V), float("nan"))
            full[sc.idx[k][:m].long().cpu()] = ar.logits_c[o0:o0 + m].float().cpu()
            logits[t] = full.view(B, S, -1)
    layer_out = []
    for i in range(eng.L):
        for k, t in enumerate(eng.tasks):
            if sc is not None and i == eng.L - 1:
                # tail-rows mode: the last layer's output exists for the rows a head reads only (NaN elsewhere)
                o0, m = sc.segs[k][0], sc.counts[k]
                full = torch.full((B * S, eng.D), float("nan"))
                full[sc.idx[k][:m].long().cpu()] = ar.xl_c[o0:o0 + m].float().cpu()
                layer_out.append(full.view(B, S, -1))
            else:
                layer_out.append(ar.xs[i + 1][k * B * S:(k + 1) * B * S].view(B, S, -1).float().cpu())
    return losses, grads, logits, layer_out
