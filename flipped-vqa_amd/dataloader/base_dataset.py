"""Per-sample construction of the flipped-VQA training inputs (the step immediately
before the hot path). Restates reference dataloader/base_dataset.py:17-173: pad/truncate the three
token streams to `max_seq_len`, derive the CE labels and masks from the prefix indices, and emit
the frame-slot index ranges. Arithmetic is integer/byte work on the host: results are bit-exact
against fixtures generated from the reference (tests/golden/loader_nextqa.npz)."""
from typing import Dict, List, Optional, Sequence, Tuple

import torch
from torch.utils.data import Dataset

TASKS = ("vqa", "vaq", "qav")


class BaseDataset(Dataset):
    def __init__(self, args, tokenizer, split):
        self.args = args
        self.max_feats = args.max_feats
        self.features_dim = 768
        self.audio_features_dim = 1024
        self.tokenizer = tokenizer
        self.max_seq_len = args.max_seq_len
        self.split = split

    def _get_padding_id(self, text_id: Sequence[torch.Tensor]) -> torch.Tensor:
        """(n_seq, max_seq_len) int64, -1 beyond each sequence, longer sequences cut
        (reference base_dataset.py:17-28, which also prints a notice on overflow)."""
        S = self.max_seq_len
        out = torch.full((len(text_id), S), -1, dtype=torch.int64)
        for row, ids in zip(out, text_id):
            n = min(len(ids), S)
            row[:n] = ids[:n]
            if len(ids) > S:
                print("max sequence length overflow")
        return out

    def _get_text_token(self, text, answer: int, options: Optional[List[str]] = None):
        """-> text_id, label, video_start, video_index, label_mask, prefix_index (dicts over vqa/vaq/qav).
        vqa/vaq labels: the token ids from the prefix index on, 0 (= ignore_index) before it and on
        padding; qav labels: -1 everywhere except [p, p+F) = 0..F-1 (clipped at the sequence end);
        text ids: padding and frame placeholders (-1 / -2) become 0 (reference base_dataset.py:30-173)."""
        tok, F = self.tokenizer, self.max_feats
        kw = dict(text=text, max_feats=F, split=self.split, answer_mapping=self.answer_mapping, answer=answer,
                  options=options)
        seqs, prefix, vstart = {}, {}, {}
        seqs["vqa"], prefix["vqa"], vstart["vqa"] = tok.encode_vqa(**kw)
        seqs["vaq"], prefix["vaq"], vstart["vaq"] = tok.encode_vaq(**kw)
        seqs["qav"], prefix["qav"] = tok.encode_qav(**kw)
        vstart["qav"] = prefix["qav"]
        padded = {t: self._get_padding_id([torch.tensor(s, dtype=torch.int64) for s in seqs[t]]) for t in TASKS}

        label, label_mask = {}, {}
        for t in ("vqa", "vaq"):
            lab = padded[t].clone()
            lab[:, :prefix[t]] = -1
            keep = lab >= 0
            label[t] = torch.where(keep, lab, torch.zeros_like(lab))
            label_mask[t] = keep.float()
        S = padded["qav"].shape[1]
        p = prefix["qav"]
        n = max(0, min(S - p, F))
        lab = torch.full_like(padded["qav"], -1)
        lab[:, p:p + n] = torch.arange(n)
        label["qav"] = lab
        m = torch.zeros_like(padded["qav"])
        m[:, p] = 1                                        # only the first frame slot (reference :84-85)
        label_mask["qav"] = m.float()

        text_id = {t: padded[t].clamp_min(0) for t in TASKS}
        video_index = {t: torch.arange(prefix[t], prefix[t] + F) for t in TASKS}
        return text_id, label, dict(vstart), video_index, label_mask, dict(prefix)
