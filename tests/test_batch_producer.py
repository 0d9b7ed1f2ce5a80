"""Batch producer against fixtures generated from the reference's own
dataloader + prompt templates (oracle/gen_golden_loader.py): integer / byte work, so everything is
compared bit-exactly. Runs on the CPU; the device staging part has its own -m gpu test."""
import json
import os
import types

import numpy as np
import pandas as pd
import pytest
import torch

import dataloader
from llama.tokenizer import Tokenizer
from oracle.fake_sp import FakeSentencePiece

GOLD = dict(np.load(os.path.join(os.path.dirname(__file__), "golden", "loader_nextqa.npz")))
TASKS = ("vqa", "vaq", "qav")
CONFIGS = [("s128_train", 128, "train", False), ("s128_val", 128, "val", False), ("s115_train", 115, "train", False),
           ("s128_gen_train", 128, "train", True), ("s128_gen_val", 128, "val", True)]


@pytest.fixture(scope="module")
def data_root(tmp_path_factory):
    root = tmp_path_factory.mktemp("data")
    rows = json.loads(str(GOLD["rows_json"]))
    cols = {"video": [r[0] for r in rows], "question": [r[1] for r in rows], "answer": [r[2] for r in rows],
            "type": [r[3] for r in rows]}
    for i in range(5):
        cols[f"a{i}"] = [r[4][i] for r in rows]
    os.makedirs(root / "nextqa" / "video_features")
    for split in ("train", "val"):
        pd.DataFrame(cols).to_csv(root / "nextqa" / f"{split}.csv", index=False)
    feats = {str(k): torch.from_numpy(GOLD[f"feat__{k}"]) for k in GOLD["feat_names"]}
    torch.save(feats, root / "nextqa" / "video_features" / "clipvitl14.pth")
    return str(root)


def make_dataset(data_root, S, split, gen):
    args = types.SimpleNamespace(max_feats=10, max_seq_len=S, dataset="nextqa", audio=False, audio_only=False,
                                 audio_merge="none", debug=False, is_generation_task=gen, synthetic=True,
                                 data_root=data_root)
    tok = Tokenizer("/nonexistent/tokenizer.model", args)
    tok.sp_model = FakeSentencePiece()
    return dataloader.NextQA(args=args, tokenizer=tok, split=split)


@pytest.mark.parametrize("name,S,split,gen", CONFIGS)
def test_samples_match_reference(data_root, name, S, split, gen):
    ds = make_dataset(data_root, S, split, gen)
    assert len(ds) == GOLD[f"{name}__video_len"].shape[0]
    samples = [ds[i] for i in range(len(ds))]
    for key in ("text_id", "label", "label_mask", "video_index"):
        for t in TASKS:
            got = torch.stack([s[key][t] for s in samples]).numpy()
            ref = GOLD[f"{name}__{key}__{t}"]
            assert got.dtype == ref.dtype and got.shape == ref.shape, (key, t)
            assert np.array_equal(got, ref), (key, t)
    for key in ("video_start", "prefix_index"):
        for t in TASKS:
            assert [s[key][t] for s in samples] == GOLD[f"{name}__{key}__{t}"].tolist(), (key, t)
    assert [s["video_len"] for s in samples] == GOLD[f"{name}__video_len"].tolist()
    assert [s["qtype"] for s in samples] == GOLD[f"{name}__qtype"].tolist()
    if f"{name}__video" in GOLD:
        assert np.array_equal(torch.stack([s["video"] for s in samples]).numpy(), GOLD[f"{name}__video"])
    b = dataloader.batch_collate(samples[:4])
    for key in ("text_id", "label", "label_mask", "video_index"):
        for t in TASKS:
            assert np.array_equal(b[key][t].numpy(), GOLD[f"{name}__batch__{key}__{t}"]), (key, t)
    for t in TASKS:
        assert list(b["video_start"][t]) == GOLD[f"{name}__batch__video_start__{t}"].tolist()
    assert np.array_equal(b["answer"].numpy(), GOLD[f"{name}__batch__answer"])
    if f"{name}__batch__video" in GOLD:
        assert np.array_equal(b["video"].numpy(), GOLD[f"{name}__batch__video"])


def test_frame_sampling_rule():
    from dataloader.nextqa import sample_frames
    x = torch.arange(25 * 4, dtype=torch.float32).view(25, 4)
    y, n = sample_frames(x, 10)
    assert n == 10 and torch.equal(y, x[[(j * 25) // 10 for j in range(10)]])
    y, n = sample_frames(x[:3], 10)
    assert n == 3 and torch.equal(y[:3], x[:3]) and float(y[3:].abs().sum()) == 0.0
    y, n = sample_frames(x[:10], 10)
    assert n == 10 and y is not None and torch.equal(y, x[:10])


def test_train_batch_feeds_the_model_contract(data_root):
    """The collated train batch has the fields and shapes llama/model.py:254-264 reads."""
    ds = make_dataset(data_root, 128, "train", False)
    b = dataloader.batch_collate([ds[i] for i in range(4)])
    assert b["video"].shape == (4, 10, 768) and b["video"].dtype == torch.float32
    for t in TASKS:
        assert b["text_id"][t].shape == (4, 1, 128) and b["text_id"][t].dtype == torch.int64
        assert b["label"][t].shape == (4, 1, 128)
        assert int(b["text_id"][t].min()) >= 0
    assert b["video_index"]["qav"].shape == (4, 10)
    vs = b["video_start"]["vqa"][0]
    assert all(int(b["text_id"]["vqa"][i, 0, vs]) == 0 for i in range(4))      # frame placeholders masked to 0


def test_unbuilt_datasets_and_audio_are_rejected(data_root):
    args = types.SimpleNamespace(dataset="tvqa", batch_size=2, num_workers=0, pin_mem=False)
    with pytest.raises(NotImplementedError):
        dataloader.load_data(args, None)
    with pytest.raises(NotImplementedError):
        a = types.SimpleNamespace(max_feats=10, max_seq_len=128, dataset="nextqa", audio=True, audio_only=False,
                                  data_root=data_root)
        dataloader.NextQA(args=a, tokenizer=None, split="train")


# ------------------------------------------------------------------ device staging (fvqa/batch_producer.py)
def _equal_batches(got, ref):
    for k, v in ref.items():
        if torch.is_tensor(v):
            assert torch.equal(got[k].cpu(), v), k
        elif isinstance(v, dict):
            for t, x in v.items():
                if torch.is_tensor(x):
                    assert torch.equal(got[k][t].cpu(), x), (k, t)
                else:
                    assert got[k][t] == x, (k, t)
        else:
            assert got[k] == v, k


def _shrink(batch, n):
    out = {}
    for k, v in batch.items():
        if torch.is_tensor(v):
            out[k] = v[:n]
        elif isinstance(v, dict):
            out[k] = {t: x[:n] for t, x in v.items()}
        else:
            out[k] = v[:n] if isinstance(v, list) else v
    return out


def _source_batches(data_root, n=6):
    ds = make_dataset(data_root, 128, "train", False)
    samples = [ds[i] for i in range(len(ds))]
    batches = [dataloader.batch_collate([samples[(3 * i + j) % len(samples)] for j in range(4)]) for i in range(n)]
    batches[-1] = _shrink(batches[-1], 2)                  # drop_last=False: a short last batch
    return batches


def test_producer_passthrough_cpu(data_root):
    from fvqa.batch_producer import DeviceBatchProducer
    src = _source_batches(data_root)
    prod = DeviceBatchProducer(src, "cpu", depth=2)
    assert len(prod) == len(src)
    seen = 0
    for got, ref in zip(prod, src):
        _equal_batches(got, ref)
        seen += 1
    assert seen == len(src)
    seen = sum(1 for _ in prod)                            # a second epoch over the same producer
    assert seen == len(src)


def test_producer_packs_in_dataloader_workers(data_root):
    """With a real DataLoader and worker processes the flattening happens in the workers (one tensor per batch
    crosses the process boundary); what the consumer sees is unchanged."""
    from fvqa.batch_producer import DeviceBatchProducer
    ds = make_dataset(data_root, 128, "train", False)
    mk = lambda workers: torch.utils.data.DataLoader(ds, batch_size=3, shuffle=False, num_workers=workers,
                                                     collate_fn=dataloader.batch_collate)
    ref = list(mk(0))
    loader = mk(2)
    prod = DeviceBatchProducer(loader, "cpu", depth=2)
    assert type(loader.collate_fn).__name__ == "_PackingCollate"
    n = 0
    for got, want in zip(prod, ref):
        _equal_batches(got, want)
        n += 1
    assert n == len(ref) == 3                               # 8 samples: 3 + 3 + 2


def test_producer_surfaces_loader_errors():
    from fvqa.batch_producer import DeviceBatchProducer

    def broken():
        yield {"video": torch.zeros(2, 10, 768), "text_id": {"vqa": torch.zeros(2, 1, 8, dtype=torch.int64)}}
        raise RuntimeError("reader failed")

    with pytest.raises(RuntimeError, match="reader failed"):
        for _ in DeviceBatchProducer(broken(), "cpu"):
            pass


@pytest.mark.gpu
def test_producer_device_views_and_training_step(data_root):
    """On the GPU: batches arrive as views of the staged device buffer, bit-identical to the source, and a
    training step fed by the producer gives the same losses as one fed by the CPU batch."""
    from fvqa import synth
    from fvqa.batch_producer import DeviceBatchProducer
    from tests.gpu_util import build_model
    src = _source_batches(data_root, n=7)
    prod = DeviceBatchProducer(src, "cuda", depth=2)
    kept = []
    for got, ref in zip(prod, src):
        assert got["video"].is_cuda and got["text_id"]["vqa"].is_cuda and isinstance(got["video_start"]["vqa"], list)
        _equal_batches(got, ref)
        kept.append({"v": got["video"].clone()})
    assert len(kept) == len(src) and prod.h2d_bytes > 0
    cfg = synth.preset("tiny", vaq=True, qav=True, max_seq_len=128, batch_size=4, vocab_size=32000)
    model, _ = build_model(cfg, torch.float32)
    ref_losses = [tuple(float(x) for x in model(b)) for b in src[:3]]
    got_losses = [tuple(float(x) for x in model(b)) for b, _ in zip(DeviceBatchProducer(src, "cuda", depth=3), range(3))]
    assert ref_losses == got_losses
