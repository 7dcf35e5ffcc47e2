"""Parameter-free text tower + tokenizer stand-ins shared by make_golden.py and the tests.

The real tower (frozen RoBERTa-base) cannot be part of a fixture (no weights offline, 125 M
parameters).  The stub keeps the interface the model uses -- ``tokenizer.batch_encode_plus(texts,
padding='longest', return_tensors='pt') -> BatchEncoding`` and ``text_encoder(**tokenized)
.last_hidden_state`` (B, L, 768), ``text_encoder.config.hidden_size`` -- with a closed-form
embedding so both sides compute identical features.
"""
import types
import zlib

import torch
from torch import nn
from transformers import BatchEncoding


class StubTokenizer:
    pad_id, bos_id, eos_id = 1, 0, 2

    def batch_encode_plus(self, texts, padding="longest", return_tensors="pt"):
        rows = [[self.bos_id] + [3 + zlib.crc32(w.encode()) % 50000 for w in t.split()] + [self.eos_id]
                for t in texts]
        width = max(len(r) for r in rows)
        ids = torch.full((len(rows), width), self.pad_id, dtype=torch.long)
        att = torch.zeros((len(rows), width), dtype=torch.long)
        for i, r in enumerate(rows):
            ids[i, :len(r)] = torch.tensor(r)
            att[i, :len(r)] = 1
        return BatchEncoding({"input_ids": ids, "attention_mask": att})


class StubTextEncoder(nn.Module):
    def __init__(self, hidden_size=768):
        super().__init__()
        self.config = types.SimpleNamespace(hidden_size=hidden_size)
        self.register_buffer("freq", torch.linspace(0.1, 3.0, hidden_size), persistent=False)

    def forward(self, input_ids=None, attention_mask=None, **_):
        pos = torch.arange(input_ids.shape[1], device=input_ids.device, dtype=torch.float32)
        x = (input_ids.float()[..., None] * 1e-3 + pos[None, :, None] * 0.37) * self.freq
        h = torch.sin(x) * attention_mask[..., None].float()
        return types.SimpleNamespace(last_hidden_state=h)


def factory():
    return StubTokenizer(), StubTextEncoder()
