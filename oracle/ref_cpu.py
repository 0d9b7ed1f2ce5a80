"""TEST INFRASTRUCTURE — CPU restatement of the Flipped-VQA training step (oracle).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this file;
the product path (flipped-vqa_amd/) never does and fails loudly without its HIP library.

What it is: the arithmetic of reference llama/model.py:31-365 (RMSNorm, RoPE, adapter-gated
attention, SwiGLU, visual projection + frame splice, the three flipped losses) written out
as explicit tensor algebra with a HAND-DERIVED backward (no autograd), in fp32 or fp64 on
torch-CPU. Each function cites the reference lines it restates. It is pinned by
tests/golden/*.npz, which oracle/gen_golden.py produced by running the reference itself in
this container (fp32-shim mode); tests/test_oracle_golden.py checks this file
against those fixtures.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional

import torch


# ------------------------------------------------------------------------------ primitives
def rmsnorm_fwd(x, w, eps):
    """reference llama/model.py:37-42 — y = x * rsqrt(mean(x^2) + eps) * w."""
    rstd = torch.rsqrt((x * x).mean(-1, keepdim=True) + eps)
    return x * rstd * w, rstd


def rmsnorm_bwd(g, x, w, rstd):
    """dx for y = x*rstd*w with w frozen."""
    gh = g * w
    D = x.shape[-1]
    return rstd * gh - x * (rstd ** 3) * ((gh * x).sum(-1, keepdim=True) / D)


def rope_tables(S, Dh, dtype, theta=10000.0):
    """reference llama/model.py:45-50 — angle(p, i) = p * theta^(-2i/Dh); returns cos, sin (S, Dh/2).

    The reference builds the table in fp32 (torch.polar of an fp32 outer product); so do we,
    then widen, so fp64 runs still see the fp32-rounded angles the reference uses.
    """
    freqs = 1.0 / (theta ** (torch.arange(0, Dh, 2)[: Dh // 2].float() / Dh))
    ang = torch.outer(torch.arange(S).float(), freqs).float()
    return torch.cos(ang).to(dtype), torch.sin(ang).to(dtype)


def rope_apply(t, cos, sin, inverse=False):
    """reference llama/model.py:61-67 — rotate adjacent pairs (2i, 2i+1); t is (N, S, H, Dh)."""
    e, o = t[..., 0::2], t[..., 1::2]
    c, s = cos[None, :, None, :], sin[None, :, None, :]
    if inverse:
        s = -s
    out = torch.empty_like(t)
    out[..., 0::2] = e * c - o * s
    out[..., 1::2] = e * s + o * c
    return out


def _text_bias(S, F, vs, g2, dtype):
    """(H, S, S) additive term: causal -inf plus gate2 on rows >= vs+F, cols [vs, vs+F)
    (reference llama/model.py:103-104,116-119,299-300)."""
    H = g2.shape[0]
    m = torch.full((S, S), float("-inf"), dtype=dtype).triu(1)
    m = m[None].repeat(H, 1, 1)
    if vs is not None and vs >= 0:
        m[:, vs + F:, vs:vs + F] += g2[:, None, None]
    return m


def attn_fwd(q, k, v, ak, av, gate1, gate2, vstart, F):
    """reference llama/model.py:98-126. q,k,v (N,S,H,Dh) with RoPE already applied;
    ak, av (A,H,Dh) adapter keys/values (no RoPE); vstart: per-sequence int (-1 = no gate2 bias).
    Returns o (N,S,H,Dh) and the cache for attn_bwd."""
    N, S, H, Dh = q.shape
    sc = 1.0 / math.sqrt(Dh)
    g1 = torch.tanh(gate1)
    o = torch.empty_like(q)
    Pa_all, Pt_all = [], []
    for n in range(N):
        qn = q[n].transpose(0, 1)                          # (H,S,Dh)
        s_a = torch.einsum("hsd,ahd->hsa", qn, ak) * sc
        s_t = torch.einsum("hsd,thd->hst", qn, k[n]) * sc
        s_t = s_t + _text_bias(S, F, int(vstart[n]), gate2, q.dtype)
        Pa = torch.softmax(s_a, -1)
        Pt = torch.softmax(s_t, -1)
        on = g1[:, None, None] * torch.einsum("hsa,ahd->hsd", Pa, av) + torch.einsum("hst,thd->hsd", Pt, v[n])
        o[n] = on.transpose(0, 1)
        Pa_all.append(Pa)
        Pt_all.append(Pt)
    return o, (torch.stack(Pa_all), torch.stack(Pt_all))


def attn_bwd(do, q, k, v, ak, av, gate1, gate2, vstart, F, cache):
    """Hand-derived backward of attn_fwd."""
    N, S, H, Dh = q.shape
    sc = 1.0 / math.sqrt(Dh)
    g1 = torch.tanh(gate1)
    Pa_all, Pt_all = cache
    dq, dk, dv = torch.zeros_like(q), torch.zeros_like(k), torch.zeros_like(v)
    dak, dav = torch.zeros_like(ak), torch.zeros_like(av)
    dg1 = torch.zeros_like(gate1)
    dg2 = torch.zeros_like(gate2)
    for n in range(N):
        don = do[n].transpose(0, 1)                        # (H,S,Dh)
        qn = q[n].transpose(0, 1)
        Pa, Pt = Pa_all[n], Pt_all[n]
        doVa = torch.einsum("hsd,ahd->hsa", don, av)
        dg1 += (doVa * Pa).sum((1, 2))
        dPa = g1[:, None, None] * doVa
        dPt = torch.einsum("hsd,thd->hst", don, v[n])
        dav += g1[None, :, None] * torch.einsum("hsa,hsd->ahd", Pa, don)
        dv[n] = torch.einsum("hst,hsd->thd", Pt, don)
        dSa = Pa * (dPa - (dPa * Pa).sum(-1, keepdim=True))
        dSt = Pt * (dPt - (dPt * Pt).sum(-1, keepdim=True))
        vs = int(vstart[n])
        if vs >= 0:
            dg2 += dSt[:, vs + F:, vs:vs + F].sum((1, 2))
        dqn = (torch.einsum("hsa,ahd->hsd", dSa, ak) + torch.einsum("hst,thd->hsd", dSt, k[n])) * sc
        dq[n] = dqn.transpose(0, 1)
        dk[n] = torch.einsum("hst,hsd->thd", dSt, qn) * sc
        dak += torch.einsum("hsa,hsd->ahd", dSa, qn) * sc
    dgate1 = dg1 * (1.0 - g1 * g1)
    return dq, dk, dv, dak, dav, dgate1, dg2


def silu(x):
    return x * torch.sigmoid(x)


def swiglu_bwd(dz, a, b):
    """z = silu(a)*b (reference llama/model.py:142)."""
    sg = torch.sigmoid(a)
    da = dz * b * sg * (1.0 + a * (1.0 - sg))
    db = dz * a * sg
    return da, db


def ce_mean(logits, labels, ignore_index):
    """torch.nn.CrossEntropyLoss(ignore_index=..) mean over valid rows (reference
    llama/model.py:233-235,350,356,361). Returns loss and dloss/dlogits."""
    valid = labels != ignore_index
    nv = int(valid.sum())
    lse = torch.logsumexp(logits, -1)
    safe = labels.clamp(min=0)
    picked = logits.gather(1, safe[:, None])[:, 0]
    per = (lse - picked) * valid
    loss = per.sum() / nv if nv > 0 else torch.tensor(float("nan"), dtype=logits.dtype)
    dl = torch.softmax(logits, -1)
    dl[torch.arange(len(labels)), safe] -= 1.0
    dl = dl * (valid[:, None].to(logits.dtype) / max(nv, 1))
    return loss, dl


# ------------------------------------------------------------------------------ whole step
class RefModel:
    """Holds the (frozen + trainable) tensors of a reference-style state dict in `dtype`."""

    def __init__(self, cfg, sd: Dict[str, torch.Tensor], dtype=torch.float32):
        self.cfg = cfg
        self.dtype = dtype
        self.sd = {k: v.to(dtype) for k, v in sd.items()}
        self.cos, self.sin = rope_tables(cfg.max_seq_len * 2, cfg.head_dim, dtype)

    def layer_ids(self):
        L = self.cfg.n_layers
        return list(range(L))[-self.cfg.adapter_layer:]     # llama/model.py:338

    # --- embedding + frame splice (llama/model.py:286-294,326-336)
    def embed(self, task, ids, labels_full, vf, vs, vindex):
        h = self.sd["tok_embeddings.weight"][ids].clone()
        if task in ("vqa", "vaq"):
            h[:, vs:vs + self.cfg.max_feats] = vf
        else:
            mask = labels_full >= 0
            h = h * (~mask)[..., None]
            h.scatter_add_(1, vindex[..., None].repeat(1, 1, self.cfg.dim), vf)
        return h

    def block_fwd(self, i, li, x, vstart):
        cfg, sd = self.cfg, self.sd
        N, S, D = x.shape
        H, Dh, A = cfg.n_heads, cfg.head_dim, cfg.adapter_len
        p = f"layers.{li}."
        xn, r1 = rmsnorm_fwd(x, sd[p + "attention_norm.weight"], cfg.norm_eps)
        q = (xn @ sd[p + "attention.wq.weight"].T).view(N, S, H, Dh)
        k = (xn @ sd[p + "attention.wk.weight"].T).view(N, S, H, Dh)
        v = (xn @ sd[p + "attention.wv.weight"].T).view(N, S, H, Dh)
        q = rope_apply(q, self.cos[:S], self.sin[:S])
        k = rope_apply(k, self.cos[:S], self.sin[:S])
        ad = sd["adapter_query.weight"][i * A:(i + 1) * A]
        ak = (ad @ sd[p + "attention.wk.weight"].T).view(A, H, Dh)
        av = (ad @ sd[p + "attention.wv.weight"].T).view(A, H, Dh)
        g1 = sd[p + "attention.gate1"].flatten()
        g2 = sd[p + "attention.gate2"].flatten()
        o, cache = attn_fwd(q, k, v, ak, av, g1, g2, vstart, cfg.max_feats)
        h = x + o.reshape(N, S, D) @ sd[p + "attention.wo.weight"].T
        hn, r2 = rmsnorm_fwd(h, sd[p + "ffn_norm.weight"], cfg.norm_eps)
        a = hn @ sd[p + "feed_forward.w1.weight"].T
        b = hn @ sd[p + "feed_forward.w3.weight"].T
        out = h + (silu(a) * b) @ sd[p + "feed_forward.w2.weight"].T
        saved = dict(x=x, r1=r1, q=q, k=k, v=v, ak=ak, av=av, o=o, cache=cache, h=h, r2=r2, a=a, b=b,
                     vstart=vstart)
        return out, saved

    def block_bwd(self, i, li, dout, sv, grads):
        cfg, sd = self.cfg, self.sd
        x = sv["x"]
        N, S, D = x.shape
        H, Dh, A = cfg.n_heads, cfg.head_dim, cfg.adapter_len
        p = f"layers.{li}."
        dz = dout @ sd[p + "feed_forward.w2.weight"]
        da, db = swiglu_bwd(dz, sv["a"], sv["b"])
        dhn = da @ sd[p + "feed_forward.w1.weight"] + db @ sd[p + "feed_forward.w3.weight"]
        dh = dout + rmsnorm_bwd(dhn, sv["h"], sd[p + "ffn_norm.weight"], sv["r2"])
        do = (dh @ sd[p + "attention.wo.weight"]).view(N, S, H, Dh)
        g1 = sd[p + "attention.gate1"].flatten()
        g2 = sd[p + "attention.gate2"].flatten()
        dq, dk, dv, dak, dav, dg1, dg2 = attn_bwd(do, sv["q"], sv["k"], sv["v"], sv["ak"], sv["av"], g1, g2,
                                                  sv["vstart"], cfg.max_feats, sv["cache"])
        dq = rope_apply(dq, self.cos[:S], self.sin[:S], inverse=True)
        dk = rope_apply(dk, self.cos[:S], self.sin[:S], inverse=True)
        dxn = (dq.reshape(N, S, D) @ sd[p + "attention.wq.weight"]
               + dk.reshape(N, S, D) @ sd[p + "attention.wk.weight"]
               + dv.reshape(N, S, D) @ sd[p + "attention.wv.weight"])
        dx = dh + rmsnorm_bwd(dxn, x, sd[p + "attention_norm.weight"], sv["r1"])
        dad = dak.reshape(A, D) @ sd[p + "attention.wk.weight"] + dav.reshape(A, D) @ sd[p + "attention.wv.weight"]
        grads["adapter_query.weight"][i * A:(i + 1) * A] += dad
        grads[p + "attention.gate1"] += dg1.view(1, H, 1, 1)
        grads[p + "attention.gate2"] += dg2.view(1, H, 1, 1)
        return dx

    def step(self, batch, loss_weights=(1.0, 1.0, 1.0), keep=False):
        """Forward + backward of reference Transformer.forward (training branch,
        llama/model.py:254-365) for d(sum_k w_k * loss_k). Returns dict(losses, grads, extras)."""
        cfg, sd, dt = self.cfg, self.sd, self.dtype
        F, D, V = cfg.max_feats, cfg.dim, cfg.vocab_size
        video = batch["video"].to(dt)
        B = video.shape[0]
        S = batch["text_id"]["vqa"].shape[-1]
        tasks = ["vqa"] + (["vaq"] if cfg.vaq else []) + (["qav"] if cfg.qav else [])
        vs = {"vqa": int(batch["video_start"]["vqa"][0]), "vaq": int(batch["video_start"]["vaq"][0]), "qav": -1}

        grads = {n: torch.zeros_like(t) for n, t in sd.items()
                 if any(m in n for m in ("gate", "adapter", "temporal_emb", "visual_proj"))}
        Wv = sd["visual_proj.weight"]
        _vf = video @ Wv.T                                            # (B,F,D)  model.py:322
        vf = _vf + sd["temporal_emb.weight"][None]                    # model.py:324
        d_vf = torch.zeros_like(vf)       # grad wrt video_feature (with temporal)
        d__vf = torch.zeros_like(vf)      # extra grad wrt _video_feature (QAV head)

        losses = {"vqa": torch.zeros((), dtype=dt), "vaq": torch.zeros((), dtype=dt), "qav": torch.zeros((), dtype=dt)}
        extras = {"logits": {}, "layer_out": {}, "final_norm": {}}
        lw = dict(zip(("vqa", "vaq", "qav"), loss_weights))
        lids = self.layer_ids()
        for task in tasks:
            ids = batch["text_id"][task].reshape(-1, S)
            lab_full = batch["label"][task].reshape(-1, S)
            lab = lab_full[:, 1:].flatten()
            vindex = batch["video_index"]["qav"]
            x = self.embed(task, ids, lab_full, vf, vs[task], vindex)
            vstart = [vs[task]] * x.shape[0]
            saved = []
            for i, li in enumerate(lids):
                x, sv = self.block_fwd(i, li, x, vstart)
                saved.append(sv)
                if keep:
                    extras["layer_out"].setdefault(task, []).append(x)
            xn, rN = rmsnorm_fwd(x, sd["norm.weight"], cfg.norm_eps)
            if keep:
                extras["final_norm"][task] = xn
            if task in ("vqa", "vaq"):
                logits = xn @ sd["output.weight"].T                 # model.py:348-350
                if keep:
                    extras["logits"][task] = logits
                loss, dl = ce_mean(logits[:, :-1].reshape(-1, V), lab, 0)
                dlog = torch.zeros_like(logits)
                dlog[:, :-1] = dl.view(x.shape[0], S - 1, V) * lw[task]
                dxn = dlog @ sd["output.weight"]
            else:
                ql = torch.einsum("nsd,nfd->nsf", xn[:, :-1], _vf) / cfg.tau     # model.py:360-361
                loss, dl = ce_mean(ql.reshape(-1, F), lab, -1)
                dl = dl.view(x.shape[0], S - 1, F) * (lw[task] / cfg.tau)
                dxn = torch.zeros_like(xn)
                dxn[:, :-1] = torch.einsum("nsf,nfd->nsd", dl, _vf)
                d__vf += torch.einsum("nsf,nsd->nfd", dl, xn[:, :-1])
            losses[task] = loss
            dx = rmsnorm_bwd(dxn, x, sd["norm.weight"], rN)
            for i in reversed(range(len(lids))):
                dx = self.block_bwd(i, lids[i], dx, saved[i], grads)
            # frame-token grads
            if task in ("vqa", "vaq"):
                d_vf += dx[:, vs[task]:vs[task] + F]
            else:
                d_vf += dx.gather(1, vindex[..., None].repeat(1, 1, D))
            del saved
        grads["temporal_emb.weight"] += d_vf.sum(0)
        dtot = d_vf + d__vf
        grads["visual_proj.weight"] += torch.einsum("bfd,bfk->dk", dtot, video)
        return dict(losses=losses, grads=grads, extras=extras, tasks=tasks)
