"""TEST INFRASTRUCTURE — a deterministic stand-in for the LLaMA SentencePiece model (no tokenizer.model
exists offline): words, punctuation marks and newlines become one token each; the four marker pieces get
the ids the reference hard-codes (llama/tokenizer.py:28-31), everything else a CRC-derived id. Used on BOTH
sides of the batch-producer parity test (the reference's prompt templates and ours), so only the layout
logic is compared, never a real vocabulary."""
import re
import zlib

MARKERS = {"Video": 15167, "Question": 16492, "Answer": 22550, "\n": 13}
_PIECE = re.compile(r"\n|\w+|[^\w\s]")


class FakeSentencePiece:
    def vocab_size(self):
        return 32000

    get_piece_size = vocab_size

    def bos_id(self):
        return 1

    def eos_id(self):
        return 2

    def pad_id(self):
        return -1

    def encode(self, s):
        out = []
        for k, piece in enumerate(_PIECE.findall(s)):
            if k == 0 and piece == "Question":       # LLaMA: a piece that opens the string carries the word-start
                out.append(894)                      # marker, "▁Question" = 894 (reference llama/model.py:519)
                continue
            if piece in MARKERS:
                out.append(MARKERS[piece])
                continue
            i = 1000 + zlib.crc32(piece.encode("utf-8")) % 28000
            while i in MARKERS.values() or i == 894:
                i += 1
            out.append(i)
        return out

    def decode(self, t):
        return ""
