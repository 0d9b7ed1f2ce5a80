"""`llama.Tokenizer` as far as the training hot path needs it.

The model constructor only reads `n_words`, `eos_id`, `a_token_id`, `q_token_id` (reference
llama/model.py:201-204, llama_vqa.py:62); the marker ids are hard-coded in the reference
(llama/tokenizer.py:28-31). With a real LLaMA `tokenizer.model` this wraps SentencePiece like the
reference; for synthetic runs (no tokenizer asset exists offline) set FVQA_SYNTHETIC_TOKENIZER=1
or pass args.synthetic=True and the LLaMA-1 constants are used.

The three flipped prompt layouts (reference llama/tokenizer.py:44-211) are restated in
`encode_vqa / encode_vaq / encode_qav`: they only need an object with
`encode(str) -> List[int]` in `self.sp_model` (SentencePiece, or any stand-in for tests) and the
hard-coded marker ids. Frame placeholders are emitted as -2 (masked to 0 by the batch producer).
"""
import os
from typing import Dict, List, Optional, Sequence, Tuple

_INSTR = {
    "vqa": "Instruction: Predict the answer based on the video and question.\n",
    "vaq": "Instruction: Predict the question based on the video and answer.\n",
    "qav": "Instruction: Predict the video based on the question and answer.\n",
}


class Tokenizer:
    LLAMA1 = dict(n_words=32000, bos_id=1, eos_id=2, pad_id=-1)

    def __init__(self, model_path: str, args=None):
        self.args = args
        self.sp_model = None
        synthetic = bool(getattr(args, "synthetic", False)) or os.environ.get("FVQA_SYNTHETIC_TOKENIZER") == "1"
        if os.path.isfile(model_path):
            from sentencepiece import SentencePieceProcessor
            self.sp_model = SentencePieceProcessor(model_file=model_path)
            self.n_words = self.sp_model.vocab_size()
            self.bos_id, self.eos_id, self.pad_id = self.sp_model.bos_id(), self.sp_model.eos_id(), self.sp_model.pad_id()
        elif synthetic:
            self.n_words = int(getattr(args, "vocab_size", 0) or self.LLAMA1["n_words"])
            self.bos_id, self.eos_id, self.pad_id = (self.LLAMA1[k] for k in ("bos_id", "eos_id", "pad_id"))
        else:
            raise AssertionError(model_path)      # the reference asserts the file exists (tokenizer.py:18)
        self.v_token_id, self.q_token_id, self.a_token_id, self.nl_id = 15167, 16492, 22550, 13

    def encode(self, s: str, bos: bool, eos: bool) -> List[int]:
        if self.sp_model is None:
            raise RuntimeError("synthetic tokenizer has no vocabulary")
        t = self.sp_model.encode(s)
        return ([self.bos_id] if bos else []) + t + ([self.eos_id] if eos else [])

    def decode(self, t: List[int]) -> str:
        return "" if self.sp_model is None else self.sp_model.decode(t)

    # ------------------------------------------------------------------ flipped prompt layouts
    def _generation(self) -> bool:
        return bool(getattr(self.args, "is_generation_task", False))

    def _answers(self, split, answer_mapping, answer, options) -> Tuple[List[str], int]:
        """Answer strings to append (one per emitted sequence) and the index of the sequence that
        defines prefix_index: train -> the gold answer only; otherwise every candidate, gold one
        locating the prefix (reference llama/tokenizer.py:68-79,91-102)."""
        cands = list(options) if self._generation() else [answer_mapping[k] for k in answer_mapping]
        if split == "train":
            return [cands[answer] if self._generation() else answer_mapping[answer]], 0
        return cands, answer

    def _enc(self, s: str) -> List[int]:
        if self.sp_model is None:
            raise RuntimeError("synthetic tokenizer has no vocabulary")
        return list(self.sp_model.encode(s))

    def encode_vqa(self, text: Optional[Dict[str, str]] = None, max_feats: int = 10, split: str = "train",
                   answer_mapping: Optional[Dict[int, str]] = None, answer: Optional[int] = None,
                   options: Optional[Sequence[str]] = None) -> Tuple[List[List[int]], int, int]:
        """[bos] Instruction…Video: | F placeholders | \n | question (+choices) Answer: The answer is <answer> [eos].
        -> (sequences, index of the first answer token, index of the first frame slot)
        (reference llama/tokenizer.py:44-103)."""
        head = [self.bos_id] + self._enc(_INSTR["vqa"] + "Video:")
        body = text["q_text"] + ("" if self._generation() else text["o_text"]) + text["a_text"]
        tails, ref = self._answers(split, answer_mapping, answer, options)
        frames = [-2] * max_feats + [self.nl_id]
        seqs = [head + frames + self._enc(body + a) + [self.eos_id] for a in tails]
        return seqs, seqs[ref].index(self.a_token_id) + 5, len(head)

    def encode_vaq(self, text: Optional[Dict[str, str]] = None, max_feats: int = 10, split: str = "train",
                   answer_mapping: Optional[Dict[int, str]] = None, answer: Optional[int] = None,
                   options: Optional[Sequence[str]] = None) -> Tuple[List[List[int]], int, int]:
        """[bos] Instruction…Video: | F placeholders | \n | (choices) Answer: The answer is <answer> \n Question: … [eos]
        -> (sequences, index of the first question token, index of the first frame slot)
        (reference llama/tokenizer.py:106-161; in generation mode the reference takes the prefix from
        sequence 0 also at validation time, :160 — kept)."""
        head = [self.bos_id] + self._enc(_INSTR["vaq"] + "Video:")
        q = text["q_text"].strip()
        body = ("\n" if self._generation() else text["o_text"]) + text["a_text"]
        tails, ref = self._answers(split, answer_mapping, answer, options)
        if self._generation():
            ref = 0
        frames = [-2] * max_feats + [self.nl_id]
        seqs = [head + frames + self._enc(body + a + "\n" + q) + [self.eos_id] for a in tails]
        return seqs, seqs[ref].index(self.q_token_id) + 2, len(head)

    def encode_qav(self, text: Optional[Dict[str, str]] = None, max_feats: int = 10, split: str = "train",
                   answer_mapping: Optional[Dict[int, str]] = None, answer: Optional[int] = None,
                   options: Optional[Sequence[str]] = None) -> Tuple[List[List[int]], int]:
        """[bos] Instruction… question (+choices) Answer: The answer is <answer> \n Video: | F placeholders | [eos]
        -> (sequences, index of the first frame slot) (reference llama/tokenizer.py:163-207)."""
        body = _INSTR["qav"] + text["q_text"] + ("" if self._generation() else text["o_text"]) + text["a_text"]
        tails, ref = self._answers(split, answer_mapping, answer, options)
        seqs = [[self.bos_id] + self._enc(body + a + "\n" + "Video:") + [-2] * max_feats + [self.eos_id]
                for a in tails]
        return seqs, seqs[ref].index(self.v_token_id) + 2
