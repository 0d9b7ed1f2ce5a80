"""The Flipped-VQA training step on MI355X: forward + backward of reference
llama/model.py:254-365 as an explicit schedule of libfvqa_hip.so kernels.

Design (MI355X-first, not a port of the reference's autograd graph):
  * the 1..3 flipped streams (vqa / vaq / qav) share every frozen weight, so they are batched
    into ONE set of n_streams*B sequences: each weight panel is read once per layer per pass and
    the projection GEMMs see M = n_streams*B*S rows (+A adapter rows that ride along);
  * frozen weights are packed once: Wq|Wk|Wv row-concatenated, W1|W3 row-interleaved in blocks of 16, plus a TRANSPOSED copy of
    every frozen matrix (288 GB of HBM3E makes 2x weights cheap) so that dX = dY·W is the same
    K-contiguous NT GEMM as the forward — one kernel family, no transposed LDS reads;
  * activations needed by the backward live in a preallocated arena (no allocator traffic);
  * all trainables (adapter queries, gates, visual projection, temporal embedding) are views of
    one flat fp32 buffer, and so are their gradients: the backward accumulates straight into the
    flat gradient buffer, which is what RCCL all-reduces and what the fused optimizer consumes.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Dict, List, Optional

import torch

from . import _lib, ops, scored

TASKS = ("vqa", "vaq", "qav")
class TailRows:
    """The rows a head reads (fvqa/scored.py), for the streams one engine runs, on its device: segment k of the compact layout =
    stream k's rows, [off_k, off_k + rows_k). LM streams come first (vqa [, vaq]), then qav. `gather` / `scatter` are the
    fvqa_row_segs of the idx / inv maps over ALL streams; `lab[k]` the shifted labels of LM segment k."""

    def __init__(self, batch: dict, tasks, device, stream_rows: int):
        idx, inv, offs = [], [], [0]
        self.lab, self.segs, self.counts = [], [], []
        mv = lambda x: x if x.device == device else x.to(device, non_blocking=True)        # noqa: E731
        for t in tasks:
            m = int(batch[scored.COUNT][t])
            rows = scored.rows_of(m)
            idx.append(mv(batch["scored_idx"][t]).reshape(-1)[:rows])
            inv.append(mv(batch["scored_inv"][t]).reshape(-1))
            if t in scored.LM_TASKS:
                self.lab.append(mv(batch["scored_lab"][t]).reshape(-1)[:rows])
            self.segs.append((offs[-1], rows))
            self.counts.append(m)
            offs.append(offs[-1] + rows)
        self.tasks = tuple(tasks)
        self.idx, self.inv = idx, inv
        self.M = offs[-1]
        self.n_lm = sum(1 for t in tasks if t in scored.LM_TASKS)
        self.M_lm = offs[self.n_lm]                                  # LM segments are the first rows of the compact matrix
        self.gather = ops.row_segs(idx, offs, stream_rows)
        self.scatter = ops.row_segs(inv, offs, stream_rows)

    def one(self, k: int, stream_rows: int):
        """(gather, scatter) segs of stream k alone, against a compact matrix that starts at its segment."""
        rows = self.segs[k][1]
        return ops.row_segs([self.idx[k]], [0, rows], stream_rows), ops.row_segs([self.inv[k]], [0, rows], stream_rows)


def stage_batch(data: dict, device) -> dict:
    """A batch dict (dataloader/__init__.py:28-90 schema, host tensors) moved to `device` for a resident-batch loop, the
    tail-row lists of its streams (fvqa/scored.py) taken while the labels are on the host and moved with the rest."""
    out = scored.annotate({k: (dict(v) if isinstance(v, dict) else v) for k, v in data.items()})
    out["video"] = out["video"].to(device)
    for k in ("text_id", "label", "video_index") + scored.FIELDS:
        if k in out:
            out[k] = {t: v.to(device) for t, v in out[k].items()}
    return out


class FrozenPack:
    """Fused + transposed device copies of the frozen LLaMA weights (storage dtype)."""

    def __init__(self, model, layer_ids: List[int]):
        self.layer_ids = layer_ids
        self.wqkv, self.wqkv_t, self.wo, self.wo_t = [], [], [], []
        self.w13, self.w13_t, self.w2, self.w2_t = [], [], [], []
        self.an, self.fn = [], []
        for li in layer_ids:
            blk = model.layers[li]
            att, ff = blk.attention, blk.feed_forward
            dt = att.wq.weight.dtype
            wqkv = torch.cat([att.wq.weight.data, att.wk.weight.data, att.wv.weight.data], 0).contiguous()
            D = att.wq.weight.shape[0]
            # parameters become views of the fused buffer: no second copy of the originals
            att.wq.weight.data = wqkv[0:D]
            att.wk.weight.data = wqkv[D:2 * D]
            att.wv.weight.data = wqkv[2 * D:3 * D]
            # W1 | W3 with rows interleaved in blocks of 16 (AB16, include/fvqa.h): the GEMM that produces a and b then
            # holds both for a hidden unit in one lane and applies SwiGLU in its epilogue. A compute copy: the module's
            # w1 / w3 parameters keep their own storage.
            Hf, Din = ff.w1.weight.shape
            w13 = torch.stack([ff.w1.weight.data.view(Hf // 16, 16, Din), ff.w3.weight.data.view(Hf // 16, 16, Din)],
                              dim=1).reshape(2 * Hf, Din).contiguous()
            self.wqkv.append(wqkv)
            self.wqkv_t.append(wqkv.t().contiguous())
            self.wo.append(att.wo.weight.data)
            self.wo_t.append(att.wo.weight.data.t().contiguous())
            self.w13.append(w13)
            self.w13_t.append(w13.t().contiguous())
            self.w2.append(ff.w2.weight.data)
            self.w2_t.append(ff.w2.weight.data.t().contiguous())
            self.an.append(blk.attention_norm.weight.data)
            self.fn.append(blk.ffn_norm.weight.data)
            assert dt == wqkv.dtype
        self.norm = model.norm.weight.data
        self.emb = model.tok_embeddings.weight.data
        self.wout = model.output.weight.data
        self.wout_t = model.output.weight.data.t().contiguous()


class Arena:
    """Activation + scratch buffers for one (n_seq, S) geometry."""

    def __init__(self, eng: "StepEngine", n_seq: int, S: int, n_lm: int):
        c = eng
        dev, dt = c.device, c.dtype
        L, D, Hf, H, A, V = c.L, c.D, c.Hf, c.H, c.A, c.V
        R = n_seq * S
        Ra = R + A
        f32 = torch.float32
        e = lambda *s, dtype=dt: torch.empty(*s, dtype=dtype, device=dev)  # noqa: E731
        self.n_seq, self.S, self.R, self.Ra = n_seq, S, R, Ra
        self.xs = e(L + 1, R, D)
        self.rstd1 = e(L, R, dtype=f32)
        self.rstd2 = e(L, R, dtype=f32)
        self.qkv = torch.zeros(L, Ra, 3 * D, dtype=dt, device=dev)   # (the q block of the adapter rows is never written)
        self.o = e(L, R, D)
        self.lse_a = e(L, n_seq * H * S, dtype=f32)
        self.lse_t = e(L, n_seq * H * S, dtype=f32)
        self.h = e(L, R, D)
        self.ab = e(L, R, 2 * Hf)                # AB16 slots; the step keeps (s, t) = (silu(a), dz/da) there, not (a, b)
        self.xn = e(R, D)
        self.adapter_c = e(L, A, D)             # storage-dtype cast of the walked layers' adapter prompts
        self.hn = e(R, D)
        self.z = e(R, Hf)
        self.xnf = e(R, D)
        self.rstdN = e(R, dtype=f32)
        self.n_lm = n_lm                        # sequences scored by the LM head (vqa [+ vaq])
        self._e, self._V = e, V
        self._logits = self._dlogits = None     # dense head (FVQA_LM_HEAD=all, parity of the logits): allocated on first use
        self._compact_cap = 0                   # scored-rows head: buffers of `cap` compact rows, grown on demand
        self.lse = e(R, dtype=f32)
        self.rowloss = e(R, dtype=f32)
        self.probs = e(R * c.F, dtype=f32)
        self.loss_sum = torch.zeros(3, 2, dtype=f32, device=dev)
        # backward scratch
        self.dxnf = e(R, D)
        self.da = e(R, D)
        self.db = e(R, D)
        self.dz = e(R, max(Hf, D))              # scratch: GEMM outputs on their way into the norm backward
        self.dab = e(R, 2 * Hf)
        self.dhn = e(R, D)
        self.dh = e(R, D)
        self.do = e(R, D)
        self.dqkv = e(Ra, 3 * D)
        self.dxn = e(R, D)
        self.d_tok = e(eng_frames(c, n_seq), D, dtype=f32)
        self.d_qav = e(eng_frames(c, n_seq), D, dtype=f32)
        self.gscale = e(3, dtype=f32)
        ws = ops.attn_bwd_workspace(n_seq, S, H, c.Dh, A)
        self.attn_ws = torch.zeros(ws, dtype=torch.uint8, device=dev)   # arrival counters start at zero


    @property
    def logits(self):
        if self._logits is None:
            self._logits = self._e(self.n_lm * self.S, self._V, dtype=torch.float32)
        return self._logits

    @property
    def dlogits(self):
        if self._dlogits is None:
            self._dlogits = self._e(self.n_lm * self.S, self._V)
        return self._dlogits

    def compact(self, M: int, D: int, Hf: int):
        """Buffers of the tail rows for M compact rows: the last layer's post-attention half, the LM head, their backward."""
        if M > self._compact_cap:
            cap = (M + 63) // 64 * 64
            e, V, f32 = self._e, self._V, torch.float32
            for n in ("og", "xg", "h_c", "hn_c", "xl_c", "xnf_c", "dxnf_c", "dcur_c", "dt_c", "dh_c", "do_c"):
                setattr(self, n, e(cap, D))
            self.ab_c, self.dab_c, self.z_c = e(cap, 2 * Hf), e(cap, 2 * Hf), e(cap, Hf)
            self.rstd2_c, self.rstdN_c = e(cap, dtype=f32), e(cap, dtype=f32)
            self.logits_c, self.dlogits_c = e(cap, V, dtype=f32), e(cap, V)
            self.lse_c, self.rowloss_c = e(cap, dtype=f32), e(cap, dtype=f32)
            self._compact_cap = cap
        return self


def eng_frames(c, n_seq):
    return (n_seq // c.n_streams) * c.F


class StepEngine:
    def __init__(self, model):
        p = model.params
        self.model = model
        self.D, self.H = p.dim, p.n_heads
        self.Dh = p.dim // p.n_heads
        self.V = model.vocab_size
        self.A, self.F = model.adapter_len, model.max_feats
        self.eps = p.norm_eps
        self.tau = float(model.tau)
        self.layer_ids = list(range(p.n_layers))[-model.adapter_layer:]
        self.L = len(self.layer_ids)
        self.Hf = model.layers[0].feed_forward.w1.weight.shape[0]
        self.tasks = ["vqa"] + (["vaq"] if model.args.vaq else []) + (["qav"] if model.args.qav else [])
        self.n_streams = len(self.tasks)
        self.device = model.tok_embeddings.weight.device
        self.dtype = model.tok_embeddings.weight.dtype
        if self.device.type != "cuda":
            raise RuntimeError("Flipped-VQA hot path runs only on a ROCm device (no CPU fallback); "
                               "move the model with model.to('cuda') first")
        if self.dtype not in (torch.float32, torch.bfloat16, torch.float16):
            raise TypeError(f"frozen weights must be float32, bfloat16 or float16, got {self.dtype}")
        self.pack = FrozenPack(model, self.layer_ids)
        cos, sin = model.rope_tables()
        self.cos = cos.to(self.device).contiguous()
        self.sin = sin.to(self.device).contiguous()
        self._arena: Dict[tuple, Arena] = {}
        self._gen_arena: Dict[tuple, Arena] = {}     # arenas of the generation path (VQA stream only)
        self._vstart: Dict[tuple, torch.Tensor] = {}
        self.saved = None
        self.last_scored = None
        self.keep_logits = False
        # "scored": the last layer's post-attention half, the heads and their backward on the rows a head reads (TailRows);
        # "all": every position, as the reference
        self.lm_head_rows = "all" if os.environ.get("FVQA_LM_HEAD", "scored") == "all" else "scored"

    # ------------------------------------------------------------------ native layer schedule
    def layer_plan(self, ar: "Arena", grads: "FlatParams", vstart: torch.Tensor):
        """fvqa_layer_plan for this arena (built once, cached on the arena; csrc/schedule.hip walks it)."""
        key = (vstart.data_ptr(), grads.flat_grad.data_ptr())
        if getattr(ar, "_plan_key", (None,))[:2] == key and ar._plan_key[2] == ops.gemm_workspace(self.device).data_ptr():
            return ar._plan
        m, pk, L = self.model, self.pack, self.L
        plan = _lib.LayerPlan()
        for k, v in dict(dtype=ops.dt_code(self.dtype), n_layers=L, n_seq=ar.n_seq, seq_len=ar.S, n_heads=self.H,
                         head_dim=self.Dh, adapter_len=self.A, max_feats=self.F, dim=self.D, hidden=self.Hf).items():
            setattr(plan, k, v)
        plan.eps = float(self.eps)
        keep = []

        def table(tensors):
            arr = (C.c_void_p * L)(*[t.data_ptr() for t in tensors])
            keep.append((arr, tensors))
            return C.cast(arr, C.POINTER(C.c_void_p))

        for name in ("wqkv", "wo", "w13", "w2", "wqkv_t", "wo_t", "w13_t", "w2_t", "an", "fn"):
            setattr(plan, name, table(getattr(pk, name)))
        gv = [m.gate_views(i) for i in range(L)]
        gg = [grads.gate_grad_views(i) for i in range(L)]
        plan.gate1, plan.gate2 = table([g[0] for g in gv]), table([g[1] for g in gv])
        plan.dgate1, plan.dgate2 = table([g[0] for g in gg]), table([g[1] for g in gg])
        plan.adapter = m.adapter_query.weight.data.data_ptr()
        plan.adapter_c = ar.adapter_c.data_ptr()
        plan.d_adapter = grads.grad_view("adapter_query.weight").data_ptr()
        plan.norm_w = pk.norm.data_ptr()
        for name in ("xs", "rstd1", "rstd2", "qkv", "o", "lse_a", "lse_t", "h", "ab", "xn", "hn", "z", "xnf", "rstdN",
                     "dz", "dab", "dh", "dqkv", "attn_ws"):
            setattr(plan, name, getattr(ar, name).data_ptr())
        plan.dcur, plan.dnxt, plan.d_o = ar.da.data_ptr(), ar.db.data_ptr(), ar.do.data_ptr()
        plan.cos_t, plan.sin_t, plan.vstart = self.cos.data_ptr(), self.sin.data_ptr(), vstart.data_ptr()
        plan.attn_ws_bytes = ar.attn_ws.numel()
        lib = _lib.load(self.dtype)
        need = int(lib.fvqa_layers_gemm_workspace(C.addressof(plan)))
        # the stream's one GEMM workspace (ops.gemm_workspace): every launch of the step, from either schedule, shares
        # its flags, slabs and ERROR word, which the loss scaler hands to the unscale kernel
        ws = ops.gemm_workspace(self.device, need)
        key = key + (ws.data_ptr(),)
        plan.gemm_ws, plan.gemm_ws_bytes = ws.data_ptr(), ws.numel()
        ar._plan, ar._plan_keep, ar._plan_key = plan, keep + [ws], key
        return plan

    @staticmethod
    def use_native_schedule() -> bool:
        # the per-kernel Python schedule stays for debugging (bench.py's launch probe lives in the library and works
        # under either schedule)
        return os.environ.get("FVQA_PY_SCHEDULE") != "1"

    # ------------------------------------------------------------------ helpers
    def arena(self, n_seq, S) -> Arena:
        key = (n_seq, S)
        if key not in self._arena:
            n_lm = (n_seq // self.n_streams) * (1 + int("vaq" in self.tasks))
            self._arena[key] = Arena(self, n_seq, S, n_lm)
        return self._arena[key]

    def vstart_tensor(self, B, vs_vqa, vs_vaq) -> torch.Tensor:
        key = (B, vs_vqa, vs_vaq)
        if key not in self._vstart:
            v = []
            for t in self.tasks:
                v += [{"vqa": vs_vqa, "vaq": vs_vaq, "qav": -1}[t]] * B
            self._vstart[key] = torch.tensor(v, dtype=torch.int32, device=self.device)
        return self._vstart[key]

    # ------------------------------------------------------------------ forward
    def check_gemm_error(self):
        """Raise if any persistent-GEMM launch of this engine reported a timed-out split-K exchange (the error word of
        its workspaces, include/fvqa.h). One small device->host read per workspace: call it at an epoch boundary."""
        bad = ops.gemm_error(device=self.device)
        if bad:
            raise RuntimeError("fvqa: a split-K exchange of the persistent GEMM timed out (workspace error word "
                               f"{bad}); the results of that step are invalid")

    def forward(self, data: dict):
        m, pk = self.model, self.pack
        dev = self.device
        F, D, V, A, H, Dh, Hf, L = self.F, self.D, self.V, self.A, self.H, self.Dh, self.Hf, self.L
        video = data["video"]
        B = video.shape[0]
        S = data["text_id"]["vqa"].shape[-1]
        n_opt = data["text_id"]["vqa"].shape[1]
        if n_opt != 1:
            raise ValueError("training path expects n_options == 1 (reference llama/model.py:267)")
        vs = {"vqa": int(data["video_start"]["vqa"][0]), "vaq": int(data["video_start"]["vaq"][0])}
        for t in ("vqa", "vaq"):
            if t in self.tasks and not (0 <= vs[t] and vs[t] + F <= S):
                raise ValueError(f"video_start[{t}]={vs[t]} does not leave room for {F} frames in S={S}")
        ids_h = {t: data["text_id"][t].reshape(B, S) for t in self.tasks}
        for t, v in ids_h.items():                              # host-side range check before the gather
            if not v.is_cuda and (int(v.min()) < 0 or int(v.max()) >= V):
                raise ValueError(f"text_id[{t}] outside [0, {V})")
        ids = {t: v.to(dev, non_blocking=True).contiguous() for t, v in ids_h.items()}
        labels = {t: data["label"][t].reshape(B, S).to(dev, non_blocking=True).contiguous() for t in self.tasks}
        video_d = video.to(dev, dtype=torch.float32, non_blocking=True).reshape(B * F, -1).contiguous()
        # (with the other host-to-device copies of the batch, not behind the layers: a pageable copy holds the host until the
        # stream reaches it)
        tl = self._tail(data, B, S) if self.lm_head_rows == "scored" else None
        qidx = None
        if "qav" in self.tasks:
            qi = data["video_index"]["qav"]
            if not qi.is_cuda and (int(qi.min()) < 0 or int(qi.max()) >= S):
                raise ValueError("video_index[qav] outside the sequence")
            qidx = qi.to(dev, non_blocking=True).contiguous()

        n_seq = self.n_streams * B
        ar = self.arena(n_seq, S)
        R, Ra = ar.R, ar.Ra
        vstart = self.vstart_tensor(B, vs["vqa"], vs["vaq"])

        # visual projection + temporal embedding (model.py:322,324)
        vf_raw = torch.empty(B * F, D, dtype=torch.float32, device=dev)
        vf_tok = torch.empty(B * F, D, dtype=self.dtype, device=dev)
        ops.visual_proj_fwd(video_d, m.visual_proj.weight.data, m.temporal_emb.weight.data, vf_raw, vf_tok)

        # embedding gather + splice per stream (model.py:286-294,326-336)
        for k, t in enumerate(self.tasks):
            h0 = ar.xs[0][k * B * S:(k + 1) * B * S]
            if t == "qav":
                ops.embed_splice(ids[t], pk.emb, vf_tok, h0, B, S, F, zero_labels=labels[t], index=qidx, mode=1)
            else:
                ops.embed_splice(ids[t], pk.emb, vf_tok, h0, B, S, F, vstart=vs[t], mode=0)

        if tl is not None:
            ar.compact(tl.M, D, Hf)
        if self.use_native_schedule():
            plan = self.layer_plan(ar, m._flat, vstart)
            self._plan_tail(plan, ar, tl)
            _lib.check(_lib.load(self.dtype).fvqa_layers_fwd(C.addressof(plan), torch.cuda.current_stream().cuda_stream),
                       "fvqa_layers_fwd")
        else:
            self._layers_fwd_py(ar, vstart, n_seq, S, tl)
        n_lm = ar.n_lm
        ar.loss_sum.zero_()
        if tl is not None:
            # the layers left the final-norm output of the tail rows in ar.xnf_c: LM segments first -> (M_lm, D) x W_out^T -> CE per
            # stream segment (one "sequence" each)
            lg = ar.logits_c[: tl.M_lm]
            ops.gemm_nt(ar.xnf_c[: tl.M_lm], pk.wout, lg)
            for k in range(tl.n_lm):
                o0, rows_k = tl.segs[k]
                seg = slice(o0, o0 + rows_k)
                ops.ce_fwd(lg[seg], tl.lab[k], ar.lse_c[seg], ar.rowloss_c[seg], ar.loss_sum[k], 1, rows_k, V, 0)
        else:
            ops.gemm_nt(ar.xnf[: n_lm * S], pk.wout, ar.logits)
        for k, t in enumerate(self.tasks):
            rows = slice(k * B * S, (k + 1) * B * S)
            if t == "qav":
                if tl is not None:      # the QAV head's kernels read dense rows (and skip the ones it ignores): its segment goes back
                    o0, rows_k = tl.segs[k]
                    ops.scatter_rows(ar.xnf_c[o0:o0 + rows_k], ar.xnf[rows], tl.one(k, B * S)[1])
                ops.qav_head_fwd(ar.xnf[rows], vf_raw, labels[t], ar.probs[k * B * S * F:], ar.rowloss[rows],
                                 ar.loss_sum[2], B, S, D, F, self.tau)
            elif tl is None:
                ops.ce_fwd(ar.logits[rows], labels[t], ar.lse[rows], ar.rowloss[rows], ar.loss_sum[k], B, S, V, 0)
        self.saved = dict(ar=ar, B=B, S=S, vs=vs, labels=labels, qidx=qidx, video=video_d, vf_raw=vf_raw,
                          vstart=vstart, tail=tl)
        self.last_scored = tl                                   # (tests: which rows the compact results belong to)
        losses = ar.loss_sum[:, 0] / ar.loss_sum[:, 1]        # mean over scored rows (NaN if none, as torch CE)
        return losses

    def _tail(self, data: dict, B: int, S: int) -> Optional[TailRows]:
        """The batch's tail-row lists on the device. They come with the batch (the batch producer and stage_batch make them
        where the labels are on the host) or are made here from host labels; None — the dense form — for device labels without
        lists: reading them back would stall the step."""
        have = data.get(scored.COUNT, {})
        if not all(t in have for t in self.tasks):
            if any(data["label"][t].is_cuda for t in self.tasks):
                return None
            data = scored.annotate({"label": {t: data["label"][t] for t in self.tasks}})
        return TailRows(data, self.tasks, self.device, B * S)

    def _plan_tail(self, plan, ar: "Arena", tl: Optional[TailRows]):
        """fvqa_layer_plan.tail for this step (rows = 0: dense)."""
        t = plan.tail
        if tl is None:
            t.rows = 0
            return
        t.rows = tl.M
        C.memmove(C.addressof(t.gather), C.addressof(tl.gather), C.sizeof(_lib.RowSegs))
        C.memmove(C.addressof(t.scatter), C.addressof(tl.scatter), C.sizeof(_lib.RowSegs))
        for f, n in (("og", "og"), ("xg", "xg"), ("h", "h_c"), ("hn", "hn_c"), ("ab", "ab_c"), ("z", "z_c"), ("xl", "xl_c"),
                     ("xnf", "xnf_c"), ("rstd2", "rstd2_c"), ("rstdN", "rstdN_c"), ("dcur", "dcur_c"), ("dab", "dab_c"),
                     ("dt", "dt_c"), ("dh", "dh_c"), ("d_o", "do_c")):
            setattr(t, f, getattr(ar, n).data_ptr())

    def _layers_fwd_py(self, ar, vstart, n_seq, S, tl=None):
        """The same layer walk as csrc/schedule.hip, one ctypes call per kernel."""
        m, pk = self.model, self.pack
        F, D, A, H, Dh, Hf, L = self.F, self.D, self.A, self.H, self.Dh, self.Hf, self.L
        R = ar.R
        adapter = m.adapter_query.weight.data.view(-1, A, D)     # (adapter_layer, A, D); model.py:304
        ops.cast_rows(adapter.reshape(L * A, D), ar.adapter_c.view(L * A, D))
        ops.rmsnorm_fwd(ar.xs[0], pk.an[0], ar.xn, ar.rstd1[0], self.eps, rows=R)
        # the K/V projections of the adapter rows (model.py:98-100) ride on a projection launch: layer i's QKV, or (kv_ahead)
        # layer i-1's W1|W3 — they depend on parameters only; layer 0's then is a launch of its own
        kv_ahead = ops.kv_rider_ahead(self.dtype)
        st = ops.swiglu_st()
        if kv_ahead:
            ops.gemm_nt(ar.adapter_c[0], pk.wqkv[0][D:], ar.qkv[0][R:, D:])
        for i in range(L):
            x = ar.xs[i]
            g1, g2 = m.gate_views(i)
            if ops.rope_in_gemm(self.dtype):                # bf16 MFMA build: q, k rotated where the projection produces them
                kv = {} if kv_ahead else dict(rider_a=ar.adapter_c[i], rider_b=pk.wqkv[i][D:], rider_out=ar.qkv[i][R:, D:])
                ops.gemm_nt_rope(ar.xn, pk.wqkv[i], ar.qkv[i][:R], (self.cos, self.sin), S, Dh, H, **kv)
                ops.attn_fwd(ar.qkv[i], ar.o[i], ar.lse_a[i], ar.lse_t[i], g1, g2, vstart, n_seq, S, H, Dh, A, F)
            else:
                ops.gemm_nt_rider(ar.xn, pk.wqkv[i], ar.qkv[i][:R], rider_a=ar.adapter_c[i], rider_b=pk.wqkv[i][D:],
                                  rider_out=ar.qkv[i][R:, D:])
                if ops.attn_rope_fused(self.dtype):         # (FVQA_ROPE_IN_GEMM=0) q,k stay raw, rotated inside
                    ops.attn_fwd(ar.qkv[i], ar.o[i], ar.lse_a[i], ar.lse_t[i], g1, g2, vstart, n_seq, S, H, Dh, A, F,
                                 rope=(self.cos, self.sin))
                else:
                    ops.rope_qk(ar.qkv[i], self.cos, self.sin, n_seq, S, H, Dh)
                    ops.attn_fwd(ar.qkv[i], ar.o[i], ar.lse_a[i], ar.lse_t[i], g1, g2, vstart, n_seq, S, H, Dh, A, F)
            if tl is not None and i == L - 1:
                # the last layer's post-attention half and the final norm on the tail rows only (csrc/schedule.hip, same calls)
                M = tl.M
                ops.gather_rows(ar.o[i], ar.og[:M], tl.gather)
                ops.gather_rows(x, ar.xg[:M], tl.gather)
                ops.gemm_nt(ar.og[:M], pk.wo[i], ar.h_c[:M], residual=ar.xg[:M])
                ops.rmsnorm_fwd(ar.h_c[:M], pk.fn[i], ar.hn_c[:M], ar.rstd2_c[:M], self.eps, rows=M)
                ops.gemm_nt_swiglu_fwd(ar.hn_c[:M], pk.w13[i], ar.ab_c[:M], ar.z_c[:M], st=st)
                ops.gemm_nt(ar.z_c[:M], pk.w2[i], ar.xl_c[:M], residual=ar.h_c[:M])
                ops.rmsnorm_fwd(ar.xl_c[:M], pk.norm, ar.xnf_c[:M], ar.rstdN_c[:M], self.eps, rows=M)
                break
            ops.gemm_nt(ar.o[i], pk.wo[i], ar.h[i], residual=x)           # h = x + o·Wo^T
            ops.rmsnorm_fwd(ar.h[i], pk.fn[i], ar.hn, ar.rstd2[i], self.eps, rows=R)
            if kv_ahead and i + 1 < L:      # z = silu(a)*b; ab[i] <- (s, t); rider: the next layer's adapter K/V rows
                ops.gemm_nt_swiglu_fwd(ar.hn, pk.w13[i], ar.ab[i], ar.z, st=True, rider_a=ar.adapter_c[i + 1],
                                       rider_b=pk.wqkv[i + 1][D:], rider_out=ar.qkv[i + 1][R:, D:])
            else:
                ops.gemm_nt_swiglu_fwd(ar.hn, pk.w13[i], ar.ab[i], ar.z, st=st)
            ops.gemm_nt(ar.z, pk.w2[i], ar.xs[i + 1], residual=ar.h[i])   # x' = h + z·W2^T
            if i + 1 < L:
                ops.rmsnorm_fwd(ar.xs[i + 1], pk.an[i + 1], ar.xn, ar.rstd1[i + 1], self.eps, rows=R)
            else:
                ops.rmsnorm_fwd(ar.xs[L], pk.norm, ar.xnf, ar.rstdN, self.eps, rows=R)

    # ------------------------------------------------------------------ backward
    def backward(self, g_losses: torch.Tensor, grads: "FlatParams"):
        """g_losses: (3,) fp32 device tensor = d(total)/d(loss_k). Accumulates into grads.flat_grad."""
        sv = self.saved
        if sv is None:
            raise RuntimeError("backward called without a saved forward")
        m, pk = self.model, self.pack
        ar: Arena = sv["ar"]
        B, S, vs, labels = sv["B"], sv["S"], sv["vs"], sv["labels"]
        F, D, V, A, H, Dh, Hf, L = self.F, self.D, self.V, self.A, self.H, self.Dh, self.Hf, self.L
        n_seq, R, Ra = ar.n_seq, ar.R, ar.Ra
        # the stream the gradients are produced on (the autograd thread's current stream): DataParallel.sync_grads orders the
        # gradient all-reduce behind it explicitly when the optimizer step runs on another stream
        grads.producer_stream = torch.cuda.current_stream(ar.gscale.device)
        ar.gscale.copy_(g_losses)
        ar.d_tok.zero_()
        has_qav = "qav" in self.tasks
        if has_qav:
            ar.d_qav.zero_()
        tl = sv.get("tail")
        for k, t in enumerate(self.tasks):
            rows = slice(k * B * S, (k + 1) * B * S)
            if t == "qav":
                ops.qav_head_bwd(ar.xnf[rows], sv["vf_raw"], labels[t], ar.probs[k * B * S * F:], ar.loss_sum[2],
                                 ar.gscale[2:3], ar.dxnf[rows], ar.d_qav, B, S, D, F, self.tau)
                if tl is not None:      # its gradient rows join the compact matrix
                    o0, rows_k = tl.segs[k]
                    ops.gather_rows(ar.dxnf[rows], ar.dxnf_c[o0:o0 + rows_k], tl.one(k, B * S)[0])
            elif tl is None:
                ops.ce_bwd(ar.logits[rows], labels[t], ar.lse[rows], ar.loss_sum[k], ar.gscale[k:k + 1],
                           ar.dlogits[rows], B, S, V, 0)
        if tl is not None:
            # d(logits) of the scored rows -> their dX rows through W_out, straight into the compact gradient of the final norm
            lg, dlg = ar.logits_c[: tl.M_lm], ar.dlogits_c[: tl.M_lm]
            for k in range(tl.n_lm):
                o0, rows_k = tl.segs[k]
                seg = slice(o0, o0 + rows_k)
                ops.ce_bwd(lg[seg], tl.lab[k], ar.lse_c[seg], ar.loss_sum[k], ar.gscale[k:k + 1], dlg[seg], 1, rows_k, V, 0)
            ops.gemm_nt(dlg, pk.wout_t, ar.dxnf_c[: tl.M_lm])
            dxnf = ar.dxnf_c
        else:
            ops.gemm_nt(ar.dlogits, pk.wout_t, ar.dxnf[: ar.n_lm * S])
            dxnf = ar.dxnf
        if self.use_native_schedule():
            plan = self.layer_plan(ar, grads, sv["vstart"])
            self._plan_tail(plan, ar, tl)
            out = C.c_void_p()
            _lib.check(_lib.load(self.dtype).fvqa_layers_bwd(C.addressof(plan), dxnf.data_ptr(), C.addressof(out),
                                                   torch.cuda.current_stream().cuda_stream), "fvqa_layers_bwd")
            cur = ar.da if out.value == ar.da.data_ptr() else ar.db
        else:
            cur = self._layers_bwd_py(ar, grads, sv, n_seq, S, tl)
        for k, t in enumerate(self.tasks):
            dh0 = cur[k * B * S:(k + 1) * B * S]
            if t == "qav":
                ops.splice_bwd(dh0, ar.d_tok, B, S, F, index=sv["qidx"], mode=1)
            else:
                ops.splice_bwd(dh0, ar.d_tok, B, S, F, vstart=vs[t], mode=0)
        ops.visual_proj_bwd(ar.d_tok, ar.d_qav if has_qav else None, sv["video"],
                            grads.grad_view("visual_proj.weight"), grads.grad_view("temporal_emb.weight"))
        self.saved = None


    def _layers_bwd_py(self, ar, grads, sv, n_seq, S, tl=None):
        m, pk = self.model, self.pack
        F, D, A, H, Dh, Hf, L = self.F, self.D, self.A, self.H, self.Dh, self.Hf, self.L
        R = ar.R
        cur, nxt = ar.da, ar.db
        t = ar.dz.view(-1)[: R * D].view(R, D)
        if tl is not None:
            M = tl.M
            ops.rmsnorm_bwd(ar.dxnf_c[:M], ar.xl_c[:M], pk.norm, ar.rstdN_c[:M], ar.dcur_c[:M], rows=M)
        else:
            ops.rmsnorm_bwd(ar.dxnf, ar.xs[L], pk.norm, ar.rstdN, cur, rows=R)
        g_adapter = grads.grad_view("adapter_query.weight").view(-1, A, D)
        st = ops.swiglu_st()                 # (FVQA_SWIGLU_AB=1 switches BOTH schedules to the a, b form)
        for i in reversed(range(L)):
            if tl is not None and i == L - 1:
                M = tl.M
                ops.gemm_nt_swiglu_bwd(ar.dcur_c[:M], pk.w2_t[i], ar.ab_c[:M], ar.dab_c[:M], st=st)
                ops.gemm_nt(ar.dab_c[:M], pk.w13_t[i], ar.dt_c[:M])
                ops.rmsnorm_bwd(ar.dt_c[:M], ar.h_c[:M], pk.fn[i], ar.rstd2_c[:M], ar.dh_c[:M], resid=ar.dcur_c[:M], rows=M)
                ops.gemm_nt(ar.dh_c[:M], pk.wo_t[i], ar.do_c[:M])
                ops.scatter_rows(ar.do_c[:M], ar.do, tl.scatter)
                ops.scatter_rows(ar.dh_c[:M], ar.dh, tl.scatter)
            else:
                if i + 1 < L:                    # dz·SwiGLU' in the epilogue; rider: the previous layer's adapter-grad rows
                    ops.gemm_nt_rider(cur, pk.w2_t[i], ar.dab, swiglu_ab=ar.ab[i], swiglu_st=st, rider_a=ar.dqkv[R:, D:],
                                      rider_b=pk.wqkv_t[i + 1][:, D:], rider_out=g_adapter[i + 1], accumulate=True)
                else:
                    ops.gemm_nt_swiglu_bwd(cur, pk.w2_t[i], ar.ab[i], ar.dab, st=st)
                ops.gemm_nt(ar.dab, pk.w13_t[i], t)
                ops.rmsnorm_bwd(t, ar.h[i], pk.fn[i], ar.rstd2[i], ar.dh, resid=cur, rows=R)
                ops.gemm_nt(ar.dh, pk.wo_t[i], ar.do)
            g1, g2 = m.gate_views(i)
            dg1, dg2 = grads.gate_grad_views(i)
            if ops.attn_rope_fused(self.dtype):
                ops.attn_bwd(ar.do, ar.qkv[i], ar.o[i], ar.lse_a[i], ar.lse_t[i], g1, g2, sv["vstart"], ar.dqkv, dg1,
                             dg2, ar.attn_ws, n_seq, S, H, Dh, A, F, rope=(self.cos, self.sin),
                             prerotated=ops.rope_in_gemm(self.dtype))
            else:
                ops.attn_bwd(ar.do, ar.qkv[i], ar.o[i], ar.lse_a[i], ar.lse_t[i], g1, g2, sv["vstart"], ar.dqkv, dg1,
                             dg2, ar.attn_ws, n_seq, S, H, Dh, A, F)
                ops.rope_qk(ar.dqkv, self.cos, self.sin, n_seq, S, H, Dh, inverse=True)
            ops.gemm_nt(ar.dqkv[:R], pk.wqkv_t[i], t)
            ops.rmsnorm_bwd(t, ar.xs[i], pk.an[i], ar.rstd1[i], nxt, resid=ar.dh, rows=R)
            cur, nxt = nxt, cur
        # layer 0's adapter-query gradient rows (+=): nothing left to ride on
        ops.gemm_nt(ar.dqkv[R:, D:], pk.wqkv_t[0][:, D:], None, tail=g_adapter[0], m_split=0)
        return cur


class FlatParams:
    """All trainables of the model as views of ONE flat fp32 buffer (and one flat grad buffer).

    Layout: adapter_query | visual_proj | temporal_emb | gates (L_all, 2, H). Parameter objects keep
    their reference names/shapes (llama_vqa.py:71-76 freeze policy), only their storage moves."""

    def __init__(self, model):
        self.model = model
        named = dict(model.named_parameters())
        self.names = ["adapter_query.weight", "visual_proj.weight", "temporal_emb.weight"]
        n_layers = len(model.layers)
        H = model.params.n_heads
        sizes = [named[n].numel() for n in self.names]
        gate_elems = n_layers * 2 * H
        total = sum(sizes) + gate_elems
        dev = named[self.names[0]].device
        self.flat = torch.empty(total, dtype=torch.float32, device=dev)
        # the gradient buffer and, behind it, the ERROR LANE: one fp32 (padded to 16 bytes) that data-parallel ranks all-reduce
        # in the same collective as the gradients — a rank whose persistent-GEMM error word is raised sets it to 1, the sum is
        # non-zero on every rank, and all replicas skip the step together (fvqa_grad_unscale_norm err_lane)
        self.grad_store = torch.zeros(total + 4, dtype=torch.float32, device=dev)
        self.flat_grad = self.grad_store[:total]
        self.err_lane = self.grad_store[total:total + 1]
        self.offsets = {}
        off = 0
        for n, sz in zip(self.names, sizes):
            p = named[n]
            self.flat[off:off + sz].copy_(p.data.reshape(-1).float())
            p.data = self.flat[off:off + sz].view(p.shape)
            self.offsets[n] = (off, sz, tuple(p.shape))
            off += sz
        self.gate_off = off
        self.H = H
        gates = self.flat[off:off + gate_elems].view(n_layers, 2, H)
        for li, blk in enumerate(model.layers):
            for j, g in enumerate((blk.attention.gate1, blk.attention.gate2)):
                gates[li, j].copy_(g.data.reshape(-1).float())
                g.data = gates[li, j].view(1, H, 1, 1)
                self.offsets[f"layers.{li}.attention.gate{j + 1}"] = (off + (li * 2 + j) * H, H, (1, H, 1, 1))
        self.gates = gates
        self.gate_grads = self.flat_grad[off:off + gate_elems].view(n_layers, 2, H)
        # per-parameter segment table for the norm-of-norms (util/misc.py:292)
        segs = sorted(v[0] for v in self.offsets.values()) + [total]
        self.seg_off = torch.tensor(segs, dtype=torch.int64, device=dev)
        self.attach_grads()

    def params(self):
        named = dict(self.model.named_parameters())
        return [named[n] for n in self.offsets]

    def grad_view(self, name):
        off, sz, shape = self.offsets[name]
        return self.flat_grad[off:off + sz].view(shape)

    def gate_grad_views(self, i):
        li = self.model.engine_layer_ids()[i]
        return self.gate_grads[li, 0], self.gate_grads[li, 1]

    def idle_offsets(self):
        """Flat offsets of the gates of layers the engine never runs (adapter_layer < n_layers): the reference
        leaves their .grad None, so optimizers must not touch them (no weight decay, no moments)."""
        live = set(self.model.engine_layer_ids())
        return [self.offsets[f"layers.{li}.attention.gate{j}"][0]
                for li in range(len(self.model.layers)) if li not in live for j in (1, 2)]

    def attach_grads(self):
        """p.grad <- view of the flat gradient buffer for every trainable."""
        named = dict(self.model.named_parameters())
        for n in self.offsets:
            p = named[n]
            if p.requires_grad:
                p.grad = self.grad_view(n)

    def grads_attached(self) -> bool:
        named = dict(self.model.named_parameters())
        for n, (off, sz, shape) in self.offsets.items():
            g = named[n].grad
            if g is None or g.data_ptr() != self.flat_grad.data_ptr() + off * 4:
                return False
        return True

    def zero_grad(self):
        self.flat_grad.zero_()
