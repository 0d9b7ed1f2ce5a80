"""NExT-QA samples (reference dataloader/nextqa.py:7-170): question/choices prompt text, CLIP frame
features sub-sampled or zero-padded to `max_feats`, and the flipped token streams of BaseDataset.
Paths follow the reference (`./data/nextqa/{split}.csv`, `./data/nextqa/video_features/clipvitl14.pth`)
under `args.data_root` (default "./data"). The audio variants are outside the hot path
 and are rejected."""
import os
from typing import Any, Dict, Tuple

import pandas as pd
import torch

from .base_dataset import BaseDataset


def sample_frames(feats: torch.Tensor, max_feats: int) -> Tuple[torch.Tensor, int]:
    """More frames than slots: take frame (j*len)//F for slot j; fewer: zero-pad
    (reference nextqa.py:71-81). -> (max_feats, dim) fp32, number of real frames."""
    n = feats.shape[0]
    if n > max_feats:
        pick = (torch.arange(max_feats) * n) // max_feats
        return feats[pick], max_feats
    if n < max_feats:
        return torch.cat([feats, feats.new_zeros(max_feats - n, feats.shape[1])], dim=0), n
    return feats, max_feats


class NextQA(BaseDataset):
    ANSWERS = {0: "(A)", 1: "(B)", 2: "(C)", 3: "(D)", 4: "(E)"}
    QTYPES = {"CH": 1, "CW": 2, "TN": 3, "TC": 4, "TP": 5, "DL": 6, "DC": 7, "DO": 8}

    def __init__(self, args: Any = None, tokenizer: Any = None, split: str = "train") -> None:
        super().__init__(args, tokenizer, split)
        if getattr(args, "audio", False) or getattr(args, "audio_only", False):
            raise NotImplementedError("audio features are outside the MI355X hot path")
        root = getattr(args, "data_root", "./data")
        self.data = pd.read_csv(os.path.join(root, "nextqa", f"{split}.csv"))
        self.answer_mapping = dict(self.ANSWERS)
        self.num_options = 5
        self.qtype_mapping = dict(self.QTYPES)
        dataset = getattr(args, "dataset", "nextqa")
        self.video_features = torch.load(os.path.join(root, dataset, "video_features", "clipvitl14.pth"))
        print(f"Num {split} data: {len(self.data)}")

    def _get_text(self, idx: int) -> Dict[str, Any]:
        row = self.data.iloc[idx]
        question = str(row["question"]).capitalize().strip()
        if not question.endswith("?"):
            question += "?"
        options = [row[f"a{i}"] for i in range(self.num_options)]
        choices = "".join(f"{self.answer_mapping[i]} {opt}\n" for i, opt in enumerate(options))
        return {"q_text": f"Question: {question}\n", "o_text": "Choices: \n" + choices,
                "a_text": "Answer: The answer is ", "options": options}

    def _get_video(self, video_id: str) -> Tuple[torch.Tensor, int]:
        if video_id in self.video_features:
            feats = self.video_features[video_id].float()
        else:
            print(video_id, "video not found!")
            feats = torch.zeros(1, self.features_dim)
        return sample_frames(feats, self.max_feats)

    def __getitem__(self, idx: int) -> Dict[str, Any]:
        row = self.data.iloc[idx]
        vid, answer = row["video"], row["answer"]
        text = self._get_text(idx)
        text_id, label, video_start, video_index, label_mask, prefix_index = self._get_text_token(
            text, answer, options=text["options"])
        video, video_len = self._get_video(f"{vid}")
        return {"vid": vid, "video": video, "video_len": video_len, "text": text, "text_id": text_id, "label": label,
                "video_start": video_start, "video_index": video_index, "label_mask": label_mask, "qid": idx,
                "answer": answer, "qtype": self.qtype_mapping[row["type"]], "prefix_index": prefix_index}

    def __len__(self) -> int:
        return len(self.data)
