#!/usr/bin/env python3
"""TEST INFRASTRUCTURE — generates tests/golden/loader_nextqa.npz by running the REFERENCE's own batch
producer (dataloader/nextqa.py, dataloader/base_dataset.py, dataloader/__init__.py:batch_collate,
llama/tokenizer.py prompt templates) in this container on a small synthetic NExT-QA-shaped table.

Shims (no reference file is edited or copied): a stub `pysrt` module (dataloader/tvqa.py:5 imports it),
torch.load -> {} for the hard-coded /scratch audio path (dataloader/nextqa.py:15-21), the reference
Tokenizer built around oracle.fake_sp.FakeSentencePiece instead of a tokenizer.model (none offline).
The fixture stores the inputs (table rows, frame features) next to the outputs, so the parity test can
rebuild the same dataset with the product's producer anywhere."""
import json
import os
import sys
import tempfile
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle.fake_sp import FakeSentencePiece  # noqa: E402

REF = "/root/reference"
TASKS = ("vqa", "vaq", "qav")

ROWS = [  # video, question, answer, type, a0..a4
    ("v3", "why is the dog running", 2, "CW", ["to play", "it is scared", "chasing a ball", "to eat", "for fun"]),
    ("v10", "how many people are there?", 0, "DC", ["two", "three", "four", "five", "six"]),
    ("v16", "what did the man do after he sat down", 4, "TN", ["stand up", "wave", "sleep", "eat", "read a book"]),
    ("v25", "where is this happening", 1, "DL", ["park", "kitchen", "beach", "school", "car"]),
    ("missing", "what is the baby holding", 3, "DO", ["spoon", "toy", "bottle", "a red ball", "phone"]),
    ("v10", "how does the lady react when the child falls down on the grass near the big old tree in the garden "
            "while the dog keeps barking loudly at the passing cars and the neighbours watch from the window", 0, "CH",
     ["she laughs and claps", "she runs to help the child up quickly", "she ignores it", "she cries", "she leaves"]),
    ("v1", "what happens before the ball is thrown", 2, "TP", ["jump", "run", "look back", "sit", "shout"]),
    ("v16", "why did the boy smile at the end", 1, "TC", ["tired", "he won", "hungry", "bored", "cold"]),
]
FRAMES = {"v1": 1, "v3": 3, "v10": 10, "v16": 16, "v25": 25}
CONFIGS = [  # name, max_seq_len, split, is_generation_task
    ("s128_train", 128, "train", False),
    ("s128_val", 128, "val", False),
    ("s115_train", 115, "train", False),        # overflow: truncated streams, VQA prefix past the end, QAV frame range clipped
    ("s128_gen_train", 128, "train", True),
    ("s128_gen_val", 128, "val", True),
]


def main():
    sys.modules.setdefault("pysrt", types.ModuleType("pysrt"))
    real_load = torch.load

    def load(path, *a, **k):
        if not os.path.exists(path):
            return {}
        k.setdefault("weights_only", False)
        return real_load(path, *a, **k)

    torch.load = load
    sys.path.insert(0, REF)
    import dataloader as D                      # the reference package
    import llama.tokenizer as T

    g = torch.Generator().manual_seed(11)
    feats = {k: torch.randn(n, 768, generator=g) for k, n in FRAMES.items()}
    tmp = tempfile.mkdtemp()
    os.makedirs(os.path.join(tmp, "data", "nextqa", "video_features"))
    import pandas as pd
    cols = {"video": [r[0] for r in ROWS], "question": [r[1] for r in ROWS], "answer": [r[2] for r in ROWS],
            "type": [r[3] for r in ROWS]}
    for i in range(5):
        cols[f"a{i}"] = [r[4][i] for r in ROWS]
    for split in ("train", "val"):
        pd.DataFrame(cols).to_csv(os.path.join(tmp, "data", "nextqa", f"{split}.csv"), index=False)
    torch.save(feats, os.path.join(tmp, "data", "nextqa", "video_features", "clipvitl14.pth"))
    os.chdir(tmp)

    out = {"rows_json": np.array(json.dumps(ROWS)), "feat_names": np.array(sorted(FRAMES))}
    for k in sorted(FRAMES):
        out[f"feat__{k}"] = feats[k].numpy()
    for name, S, split, gen in CONFIGS:
        args = types.SimpleNamespace(max_feats=10, max_seq_len=S, dataset="nextqa", audio=False, audio_only=False,
                                     audio_merge="none", debug=False, is_generation_task=gen)
        tok = object.__new__(T.Tokenizer)        # the reference class without its tokenizer.model assert
        tok.args = args
        tok.sp_model = FakeSentencePiece()
        tok.n_words, tok.bos_id, tok.eos_id, tok.pad_id = 32000, 1, 2, -1
        tok.v_token_id, tok.q_token_id, tok.a_token_id, tok.nl_id = 15167, 16492, 22550, 13
        ds = D.NextQA(args=args, tokenizer=tok, split=split)
        samples = [ds[i] for i in range(len(ds))]
        for key in ("text_id", "label", "label_mask", "video_index"):
            for t in TASKS:
                out[f"{name}__{key}__{t}"] = torch.stack([s[key][t] for s in samples]).numpy()
        for key in ("video_start", "prefix_index"):
            for t in TASKS:
                out[f"{name}__{key}__{t}"] = np.array([s[key][t] for s in samples], dtype=np.int64)
        if name == CONFIGS[0][0]:                 # frame sampling does not depend on the text configuration
            out[f"{name}__video"] = torch.stack([s["video"] for s in samples]).numpy()
        out[f"{name}__video_len"] = np.array([s["video_len"] for s in samples], dtype=np.int64)
        out[f"{name}__qtype"] = np.array([s["qtype"] for s in samples], dtype=np.int64)
        b = D.batch_collate(samples[:4])
        for key in ("text_id", "label", "label_mask", "video_index"):
            for t in TASKS:
                out[f"{name}__batch__{key}__{t}"] = b[key][t].numpy()
        for t in TASKS:
            out[f"{name}__batch__video_start__{t}"] = np.array(b["video_start"][t], dtype=np.int64)
        if name == CONFIGS[0][0]:
            out[f"{name}__batch__video"] = b["video"].numpy()
        out[f"{name}__batch__answer"] = b["answer"].numpy()
        print(name, "ok", {t: tuple(out[f"{name}__text_id__{t}"].shape) for t in TASKS}, flush=True)
    path = os.path.join(ROOT, "tests", "golden", "loader_nextqa.npz")
    np.savez_compressed(path, **out)
    print("->", path, os.path.getsize(path) // 1024, "KiB")


if __name__ == "__main__":
    main()
