from .model import ModelArgs, Transformer
from .tokenizer import Tokenizer

__all__ = ["ModelArgs", "Transformer", "Tokenizer"]
