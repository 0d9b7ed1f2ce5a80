"""Batch producer in front of the hot path: `load_data` / `batch_collate` with the reference's names
and collate schema (reference dataloader/__init__.py:15-90). Only NExT-QA is built (the dataset the
BASELINE configs are quoted on); the other readers of the reference are host-side file formats
outside SURVEY §8."""
import torch

from util import misc

from .base_dataset import TASKS, BaseDataset   # noqa: F401
from .nextqa import NextQA

dataset_mapping = {"nextqa": NextQA}
num_options_mapping = {"nextqa": 5}


def _worker_init(worker_id):
    """Loader workers yield to the training process: it is the one thread per rank whose stalls idle a GPU (a descheduled
    launch thread costs device time; a late batch only matters after `depth` batches of slack). Lower priority (best effort:
    os.nice needs no privilege to go down) and one intra-op thread each."""
    import os
    try:
        os.nice(int(os.environ.get("FVQA_LOADER_NICE", "10")))
    except OSError:
        pass
    torch.set_num_threads(1)


def load_data(args, tokenizer, split="train"):
    if args.dataset not in dataset_mapping:
        raise NotImplementedError(f"dataset {args.dataset!r}: only {sorted(dataset_mapping)} are built")
    args.num_options = num_options_mapping[args.dataset]
    dataset = dataset_mapping[args.dataset](args=args, tokenizer=tokenizer, split=split)
    sampler = torch.utils.data.DistributedSampler(dataset, num_replicas=misc.get_world_size(),
                                                  rank=misc.get_rank(), shuffle=split == "train")
    return torch.utils.data.DataLoader(dataset, sampler=sampler, batch_size=args.batch_size,
                                       num_workers=args.num_workers, collate_fn=batch_collate,
                                       pin_memory=args.pin_mem, drop_last=False,
                                       worker_init_fn=_worker_init if args.num_workers > 0 else None,
                                       persistent_workers=args.num_workers > 0 and split == "train")


def batch_collate(batch):
    """list of samples -> batch dict: tensors stacked on a new leading batch dim, Python ints / strings
    kept as lists (video_start, prefix_index, vid, text, qid) — the layout llama/model.py:254-264 reads."""
    def stack(key, task):
        return torch.stack([s[key][task] for s in batch])

    def gather(key, task):
        return [s[key][task] for s in batch]

    out = {"vid": [s["vid"] for s in batch]}
    if "video" in batch[0]:
        out["video"] = torch.stack([s["video"] for s in batch])
        out["video_len"] = torch.tensor([s["video_len"] for s in batch], dtype=torch.long)
    out["text"] = [s["text"] for s in batch]
    for key in ("text_id", "label", "video_index", "label_mask"):
        out[key] = {t: stack(key, t) for t in TASKS}
    for key in ("video_start", "prefix_index"):
        out[key] = {t: gather(key, t) for t in TASKS}
    out["qid"] = [s["qid"] for s in batch]
    out["answer"] = torch.tensor([s["answer"] for s in batch])
    out["qtype"] = torch.tensor([s["qtype"] for s in batch])
    return out
