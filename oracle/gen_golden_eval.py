#!/usr/bin/env python3
"""TEST INFRASTRUCTURE — generates tests/golden/eval_tiny.npz (run as `gen_golden_eval.py 7b_l2`: eval_7b_l2.npz; with a second
argument `peaked`: eval_<preset>_peaked.npz, from SynthConfig.peaked weights — LM head tied to the token embeddings, so every
greedy step's top-2 margin is a large fraction of the logit range and the bf16 build's token ids are pinned too) by running the
REFERENCE's generation/eval path
(llama/model.py:367-546 `Transformer.inference`) in this container: fp32-shim reference model with closed-form
weights (as oracle/gen_golden.py), validation batches built by the reference's own NExT-QA reader + prompt
templates on the synthetic table of oracle/gen_golden_loader.py (regex stand-in vocabulary, oracle/fake_sp.py).
Stored: the batch (ids, labels, prefix / video-start indices, answers, frame features), the token ids after the
31 greedy steps, the nearest-choice indices and the cosine similarities."""
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "flipped-vqa_amd"))
from fvqa import synth  # noqa: E402
import oracle.gen_golden as G  # noqa: E402
import oracle.gen_golden_loader as GL  # noqa: E402
from oracle.fake_sp import FakeSentencePiece  # noqa: E402


def main():
    M = G.install_shims()                       # Tensor.cuda / half shims + stub tokenizer class for the model
    sys.modules.setdefault("pysrt", types.ModuleType("pysrt"))
    real_load = torch.load
    torch.load = lambda p, *a, **k: real_load(p, *a, **{**k, "weights_only": False}) if os.path.exists(p) else {}
    import dataloader as D
    import llama.tokenizer as T
    import pandas as pd
    import tempfile

    # the validation batch, from the reference's reader
    g = torch.Generator().manual_seed(11)
    feats = {k: torch.randn(n, 768, generator=g) for k, n in GL.FRAMES.items()}
    tmp = tempfile.mkdtemp()
    os.makedirs(os.path.join(tmp, "data", "nextqa", "video_features"))
    rows = [r for r in GL.ROWS if len(r[1]) < 80]            # the very long question would run past S during decode
    cols = {"video": [r[0] for r in rows], "question": [r[1] for r in rows], "answer": [r[2] for r in rows],
            "type": [r[3] for r in rows]}
    for i in range(5):
        cols[f"a{i}"] = [r[4][i] for r in rows]
    pd.DataFrame(cols).to_csv(os.path.join(tmp, "data", "nextqa", "val.csv"), index=False)
    torch.save(feats, os.path.join(tmp, "data", "nextqa", "video_features", "clipvitl14.pth"))
    os.chdir(tmp)
    largs = types.SimpleNamespace(max_feats=10, max_seq_len=128, dataset="nextqa", audio=False, audio_only=False,
                                  audio_merge="none", debug=False, is_generation_task=True)
    tok = object.__new__(T.Tokenizer)
    tok.args, tok.sp_model = largs, FakeSentencePiece()
    tok.n_words, tok.bos_id, tok.eos_id, tok.pad_id = 32000, 1, 2, -1
    tok.v_token_id, tok.q_token_id, tok.a_token_id, tok.nl_id = 15167, 16492, 22550, 13
    ds = D.NextQA(args=largs, tokenizer=tok, split="val")
    batch = D.batch_collate([ds[i] for i in range(4)])

    # the reference model (tiny width — or, with argument "7b_l2", 7B width two layers deep —, full vocabulary so that the
    # prompt ids are valid)
    pname = sys.argv[1] if len(sys.argv) > 1 else "tiny"
    mode = sys.argv[2] if len(sys.argv) > 2 else ""
    assert mode in ("", "peaked", "peakedperm"), mode
    peaked = mode == "peaked"
    # "peakedperm" (round 5): LM head tied to a PERMUTATION of the embedding rows (SynthConfig.peaked_perm) — the greedy trajectory
    # moves, token t is followed by pi(t), so the ids pin the KV cache / position handling of a token loop, not only its head
    cfg = synth.preset(pname, vaq=False, qav=False, vocab_size=32000, max_seq_len=128, batch_size=4, peaked=peaked,
                       peaked_perm=mode == "peakedperm")
    model, margs = G.build_reference(M, cfg)
    margs.is_generation_task = True
    model.eval()
    model.tokenizer.decode = lambda t: ""
    model.answer_token_id = getattr(model, "answer_token_id", 22550)
    sims_seen = []
    orig = model.find_most_similar

    def spy(o, c):
        idx, sims = orig(o, c)
        sims_seen.append(sims.detach().float())
        return idx, sims

    model.find_most_similar = spy
    ids_seen = []
    orig_filter = model.filter_and_process_output_tokens

    def spy_filter(vqa_ids, mask):
        ids_seen.append(vqa_ids.detach().clone())
        return orig_filter(vqa_ids, mask)

    model.filter_and_process_output_tokens = spy_filter
    # per greedy step: the top-2 margin of the logits the step takes its argmax from, relative to their range (a record of
    # how decided the reference's choices are; tests read it to decide whether a bf16 build can be held to the ids)
    margins = []
    orig_output = model.output.forward

    def spy_output(x):
        y = orig_output(x)
        z = y.detach().float()
        if z.dim() == 3:
            z = z.reshape(-1, z.shape[-1])
        top2 = z.topk(2, dim=-1).values
        margins.append(((top2[:, 0] - top2[:, 1]) / (z.max(dim=-1).values - z.min(dim=-1).values).clamp_min(1e-30)).min().item())
        return y

    model.output.forward = spy_output
    # the reference writes the generated tokens into the batch's own tensor (vqa_id is a view of it): keep a copy
    original = {"text_id_vqa": batch["text_id"]["vqa"].clone(), "label_vqa": batch["label"]["vqa"].clone()}
    with torch.no_grad():
        best, extracted = model(batch, inference=True)
    out = {"best": best.numpy().astype(np.int64), "sims": sims_seen[0].numpy(), "ids_after": ids_seen[0].numpy(),
           "text_id_vqa": original["text_id_vqa"].numpy(), "label_vqa": original["label_vqa"].numpy(),
           "prefix_vqa": np.array(batch["prefix_index"]["vqa"], dtype=np.int64),
           "vstart_vqa": np.array(batch["video_start"]["vqa"], dtype=np.int64),
           "answer": batch["answer"].numpy(), "video": batch["video"].numpy(),
           "qtype": batch["qtype"].numpy(), "min_margin_per_call": np.array(margins, dtype=np.float32)}
    path = os.path.join(ROOT, "tests", "golden", f"eval_{pname}{'_' + mode if mode else ''}.npz")
    np.savez_compressed(path, **out)
    print("best", out["best"], "answers", out["answer"], "->", path, os.path.getsize(path) // 1024, "KiB")
    print("generated (first sample):", out["ids_after"][0, out["prefix_vqa"][0] - 2: out["prefix_vqa"][0] + 8])
    print("distinct ids per row:", [len(set(r.tolist())) for r in out["ids_after"]], "min margin", float(np.min(margins)))


if __name__ == "__main__":
    main()
