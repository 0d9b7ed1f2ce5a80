"""Which rows of an LM stream (vqa, vaq) the cross-entropy scores — host-side lists, made wherever the labels are still on the host.

Reference llama/model.py:348-350: `output(h)` at every position, `[:, :-1]` against `label[:, 1:]`, `ignore_index=0` — row (n, t)
of a stream is scored iff t <= S-2 and label[n, t+1] > 0. A row that is not scored contributes to no loss and to no gradient, so the
step (fvqa/step.py) gathers the scored rows of the final-norm output, runs the head, the cross-entropy and the head's dX on those,
and scatters the gradient rows back ("consciously fixed"; FVQA_LM_HEAD=all keeps the dense form).

Per stream, three tensors of the batch's own (B, S) shape — so that they travel like every other field of the batch dict (a
`.to(device)`, or packed into the producer's one staging buffer, fvqa/batch_producer.py) — and one host integer:

  scored_idx[t]  int32  flat[:rows]  row n*S + t of the stream that compact row j is gathered from (pad rows gather row 0)
  scored_inv[t]  int32  flat[n*S+t]  compact row of that row, -1 = not scored (its gradient row is zero)
  scored_lab[t]  int64  flat[:rows]  labels shifted by one: flat[0] = 0, flat[j + 1] = label of compact row j, 0 under pad rows —
                                     the CE kernels (label of row j = labels[j + 1], last row never scored) then see the segment
                                     as ONE sequence of `rows` positions
  scored_count[t] int                M = number of scored rows; rows = M + max(1, 2 - M) (>= 2: a "sequence" the kernels accept)

A label >= vocab_size is listed too: the CE kernels skip it exactly as the dense form does (heads.hip ce_fwd_k), it only costs a row.
"""
from __future__ import annotations

from typing import Tuple

import torch

LM_TASKS = ("vqa", "vaq")
FIELDS = ("scored_idx", "scored_inv", "scored_lab")
COUNT = "scored_count"


def rows_of(count: int) -> int:
    return count + max(1, 2 - count)


def lists_of(label: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor, int]:
    """label: host int64 (B, 1, S) or (B, S) -> (idx (B, S) int32, inv (B, S) int32, lab (B, S) int64, M)."""
    B, S = label.shape[0], label.shape[-1]
    lab = label.reshape(B, S)
    nxt = lab[:, 1:]
    ok = nxt > 0
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
    """The batch dict with the lists of every LM stream whose labels are host tensors added (in place; a no-op when they are
    there already or when the labels live on a device)."""
    labels = batch.get("label", {})
    for t in LM_TASKS:
        v = labels.get(t)
        if v is None or not torch.is_tensor(v) or v.is_cuda or t in batch.get(COUNT, {}):
            continue
        idx, inv, lab, m = lists_of(v)
        for f, x in zip(FIELDS, (idx, inv, lab)):
            batch.setdefault(f, {})[t] = x
        batch.setdefault(COUNT, {})[t] = m
    return batch
