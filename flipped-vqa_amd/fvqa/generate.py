"""Validation / generation path (reference llama/model.py:367-546, engine.py:59-145).

The reference decodes greedily by re-running the WHOLE sequence through all layers once per generated token
and per sample (31 forwards of (1, S) per sample, no KV cache). Here the prompt is run once for the whole
batch (the training forward, which already keeps every layer's q/k/v in the arena = the KV cache); each
further token recomputes only its own row per sample: RMSNorm -> QKV row -> the one-query-row gated attention
kernel over the cached keys (RoPE inside; the row's k, v join the cache) -> WO -> SwiGLU MLP -> LM head row -> argmax. Causality makes this identical to
the reference's re-forward: rows before the new token do not change, rows after it are never read.
The answer is then matched to the choices by cosine similarity of mean token embeddings, with the
reference's quirks kept (choices padded with id 0 before averaging, llama/model.py:566-575)."""
from __future__ import annotations

from typing import List, Tuple

import torch

from . import ops

N_NEW = 31          # reference llama/model.py:438: positions prefix-1 .. prefix+29


@torch.no_grad()
def greedy_decode(eng, data: dict, n_new: int = N_NEW) -> torch.Tensor:
    """-> ids (B, S) int64 on the device: option 0's prompt with the generated tokens written from
    position prefix_index['vqa'][b] on (reference llama/model.py:428-470)."""
    m, pk = eng.model, eng.pack
    dev = eng.device
    if "prefix_index" not in data or "vqa" not in data["prefix_index"]:
        raise ValueError("generation needs data['prefix_index']['vqa'] (where each answer starts; "
                         "reference llama/model.py:381)")
    ids_all = data["text_id"]["vqa"]
    B, _, S = ids_all.shape
    sub = {"video": data["video"],
           "text_id": {"vqa": ids_all[:, 0:1]}, "label": {"vqa": data["label"]["vqa"][:, 0:1]},
           "video_start": {"vqa": data["video_start"]["vqa"], "vaq": data["video_start"]["vqa"]}}
    saved = (eng.tasks, eng.n_streams, eng._arena, eng._vstart, eng.lm_head_rows)
    eng.tasks, eng.n_streams, eng._arena, eng._vstart = ["vqa"], 1, eng._gen_arena, {}
    eng.lm_head_rows = "all"                                # the prefill's logits are read at the prefix positions: every row
    try:
        eng.forward(sub)                                    # prefill: logits of every position + KV of every layer
        ar = eng.arena(B, S)
        vstart = eng.saved["vstart"]
        D, H, Dh, Hf, A, F, L, V = eng.D, eng.H, eng.Dh, eng.Hf, eng.A, eng.F, eng.L, eng.V
        fused = ops.attn_rope_fused(eng.dtype) and not ops.rope_in_gemm(eng.dtype)   # the prefill left RAW keys in the cache
        ids = ids_all[:, 0].to(dev).clone()
        prefix = torch.as_tensor([int(p) for p in data["prefix_index"]["vqa"]], device=dev)
        pos = prefix - 1                                    # start_idx of the first iteration
        logits = ar.logits.view(B, S, V)
        pred = logits[torch.arange(B, device=dev), pos.clamp(0, S - 1)].argmax(-1)
        e = lambda *s, dtype=eng.dtype: torch.empty(*s, dtype=dtype, device=dev)  # noqa: E731
        xn, hn, h, x2, o_row = e(B, D), e(B, D), e(B, D), e(B, D), e(B, D)
        qkv_row, ab, z = e(B, 3 * D), e(B, 2 * Hf), e(B, Hf)
        lg = e(B, V, dtype=torch.float32)
        for _ in range(n_new):
            ok = pos + 1 < S                                # the reference would index past the end here
            tgt = (pos + 1).clamp(max=S - 1)
            ids[torch.arange(B, device=dev), tgt] = torch.where(ok, pred, ids[torch.arange(B, device=dev), tgt])
            pos = tgt
            x = pk.emb[ids[torch.arange(B, device=dev), pos]].contiguous()
            for i in range(L):                                # the per-kernel sequence of a layer
                ops.rmsnorm_fwd(x, pk.an[i], xn, None, eng.eps, rows=B)
                ops.gemm_nt(xn, pk.wqkv[i], qkv_row)
                g1, g2 = m.gate_views(i)
                # the new row against the cached keys / values (+ adapter prefix); its k, v join the cache. The cache holds
                # ROTATED keys (fp32 build; bf16 build with RoPE in the QKV epilogue, the default) unless the bf16 build runs
                # with FVQA_ROPE_IN_GEMM=0 (raw keys, rotated on the fly): cache_rotated = !attn_rope_fused || rope_in_gemm
                ops.attn_decode(qkv_row, ar.qkv[i], o_row, g1, g2, vstart, pos, (eng.cos, eng.sin), B, S, H, Dh, A, F,
                                cache_rotated=not fused)
                ops.gemm_nt(o_row, pk.wo[i], h, residual=x)
                ops.rmsnorm_fwd(h, pk.fn[i], hn, None, eng.eps, rows=B)
                ops.gemm_nt(hn, pk.w13[i], ab)
                ops.swiglu_fwd(ab, z, B, Hf)
                ops.gemm_nt(z, pk.w2[i], x2, residual=h)
                x, x2 = x2, x
            ops.rmsnorm_fwd(x, pk.norm, xn, None, eng.eps, rows=B)
            ops.gemm_nt(xn, pk.wout, lg)
            pred = lg.argmax(-1)
        return ids
    finally:
        eng._gen_arena = eng._arena
        eng.tasks, eng.n_streams, eng._arena, eng._vstart, eng.lm_head_rows = saved


@torch.no_grad()
def match_answers(model, data: dict, ids: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor, List[dict]]:
    """-> (index of the most similar choice (B,), similarities (B, n_options), extracted answers)
    (reference llama/model.py:476-546 and helpers :551-623)."""
    tok = model.tokenizer
    emb = model.tok_embeddings.weight.data
    ids_all = data["text_id"]["vqa"].to(ids.device)
    label0 = data["label"]["vqa"][:, 0].to(ids.device)
    B, n_opt, S = ids_all.shape
    a_id, eos = tok.a_token_id, tok.eos_id
    out_vec, choice_vec, extracted = [], [], []
    for b in range(B):
        gen = ids[b, 1:][label0[b, 1:] != 0]                # generated tokens at the gold answer's positions
        hit = (gen == eos).nonzero()
        if hit.numel():
            gen = gen[: int(hit[0])]
        out_vec.append(emb[gen].float().mean(0) if gen.numel() else torch.zeros(emb.shape[1], device=ids.device))
        row0 = ids_all[b, 0].tolist()
        start = row0.index(a_id) + 5
        answers = []
        for c in range(n_opt):
            tail = ids_all[b, c, start:].tolist()
            end = start + tail.index(eos) if eos in tail else S
            answers.append(ids_all[b, c, start:end])
        padded = torch.nn.utils.rnn.pad_sequence(answers, batch_first=True, padding_value=0)
        choice_vec.append(emb[padded].float().mean(1))      # the padding id 0 is averaged in, as in the reference
        seq = ids[b].tolist()
        try:
            q0 = seq.index(894) + 2                          # "Question" at a line start (reference :519)
        except ValueError:
            q0 = 0
        q1 = seq.index(a_id)
        ans = seq[q1 + 5:]
        stop = ans.index(eos) if eos in ans else next((k for k, t in enumerate(ans) if t == 0), len(ans))
        extracted.append({"video_id": data["vid"][b] if "vid" in data else None,
                          "question": tok.decode(seq[q0:q1]), "generated_answer": tok.decode(ans[:stop])})
    o = torch.nn.functional.normalize(torch.stack(out_vec), p=2, dim=1)
    c = torch.nn.functional.normalize(torch.stack(choice_vec), p=2, dim=2)
    sims = torch.bmm(c, o.unsqueeze(-1)).squeeze(-1)
    return sims.argmax(1), sims, extracted
