"""`llama.Tokenizer` as far as the training hot path needs it.

The model constructor only reads `n_words`, `eos_id`, `a_token_id`, `q_token_id` (reference
llama/model.py:201-204, llama_vqa.py:62); the marker ids are hard-coded in the reference
(llama/tokenizer.py:28-31). With a real LLaMA `tokenizer.model` this wraps SentencePiece like the
reference; for synthetic runs (no tokenizer asset exists offline) set FVQA_SYNTHETIC_TOKENIZER=1
or pass args.synthetic=True and the LLaMA-1 constants are used. The three flipped prompt
templates (reference llama/tokenizer.py:44-211) are host-side string work outside this path.
"""
import os
from typing import List


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
