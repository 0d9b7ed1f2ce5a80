"""Which rows of a stream a head reads ("tail rows") — host-side lists, made wherever the labels are still on the host.

Reference llama/model.py:348-350: `output(h)` at every position, `[:, :-1]` against `label[:, 1:]`, `ignore_index=0` — row (n, t)
of an LM stream (vqa, vaq) is scored iff t <= S-2 and label[n, t+1] > 0; the QAV head (:359-361, `ignore_index=-1`) reads row
(n, t) iff label[n, t+1] >= 0 (the frame-token positions). A row no head reads contributes to no loss and to no gradient — and
nothing else consumes the LAST layer's output — so the step (fvqa/step.py, csrc/schedule.hip) gathers those rows after the last
layer's attention and runs WO + residual, the FFN, the final norm, the heads and all of their backward on them alone, scattering
the gradient rows back in front of the attention backward ("consciously fixed"; FVQA_LM_HEAD=all keeps the
dense form).

Per stream, three tensors of the batch's own (B, S) shape — so that they travel like every other field of the batch dict (a
`.to(device)`, or packed into the producer's one staging buffer, fvqa/batch_producer.py) — and one host integer:

  scored_idx[t]  int32  flat[:rows]  row n*S + t of the stream that compact row j is gathered from (pad rows gather row 0)
  scored_inv[t]  int32  flat[n*S+t]  compact row of that row, -1 = not scored (its gradient row is zero)
  scored_lab[t]  int64  flat[:rows]  labels shifted by one: flat[0] = 0, flat[j + 1] = label of compact row j, 0 under pad rows —
                                     the CE kernels (label of row j = labels[j + 1], last row never scored) then see the segment
                                     as ONE sequence of `rows` positions
  scored_count[t] int                M = number of scored rows; rows = M + max(1, 2 - M) (>= 2: a "sequence" the kernels accept)
For the qav stream the same idx / inv / count (its head's kernels read dense rows and their own labels; scored_lab["qav"] is unused).

A label >= vocab_size is listed too: the CE kernels skip it exactly as the dense form does (heads.hip ce_fwd_k), it only costs a row.
"""
from __future__ import annotations

from typing import Tuple

import torch

LM_TASKS = ("vqa", "vaq")
TASKS = ("vqa", "vaq", "qav")
FIELDS = ("scored_idx", "scored_inv", "scored_lab")
COUNT = "scored_count"


def rows_of(count: int) -> int:
    return count + max(1, 2 - count)


def lists_of(label: torch.Tensor, qav: bool = False) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor, int]:
    """label: host int64 (B, 1, S) or (B, S) -> (idx (B, S) int32, inv (B, S) int32, lab (B, S) int64, M).
    qav: the QAV head's rule (next label >= 0: a frame index) instead of the LM heads' (next label > 0)."""
    B, S = label.shape[0], label.shape[-1]
    lab = label.reshape(B, S)
    nxt = lab[:, 1:]
    ok = (nxt >= 0) if qav else (nxt > 0)
    n, t = torch.nonzero(ok, as_tuple=True)                  # row-major: (n, t) order
    rows = (n * S + t).to(torch.int32)
    m = int(rows.numel())
    idx = torch.zeros(B * S, dtype=torch.int32)
    inv = torch.full((B * S,), -1, dtype=torch.int32)
    out = torch.zeros(B * S, dtype=torch.int64)
    idx[:m] = rows
    inv[rows.long()] = torch.arange(m, dtype=torch.int32)
    out[1:m + 1] = nxt[ok]
    return idx.view(B, S), inv.view(B, S), out.view(B, S), m


def annotate(batch: dict) -> dict:
    """The batch dict with the lists of every stream whose labels are host tensors added (in place; a no-op when they are
    there already or when the labels live on a device)."""
    labels = batch.get("label", {})
    for t in TASKS:
        v = labels.get(t)
        if v is None or not torch.is_tensor(v) or v.is_cuda or t in batch.get(COUNT, {}):
            continue
        idx, inv, lab, m = lists_of(v, qav=(t == "qav"))
        for f, x in zip(FIELDS, (idx, inv, lab)):
            batch.setdefault(f, {})[t] = x
        batch.setdefault(COUNT, {})[t] = m
    return batch
