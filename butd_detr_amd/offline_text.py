"""Text tower for boxes without network access.

The reference builds ``RobertaTokenizerFast/RobertaModel.from_pretrained('roberta-base')``
(models/bdetr.py:72-77).  Offline there are no weights and transformers>=5 cannot build the fast
tokenizer without its vocab files, so benchmarks inject:

* ``random_roberta_base()`` -- a RANDOM-INIT ``RobertaModel`` with the roberta-base architecture
  (12 layers, hidden 768, 12 heads, vocab 50265, 514 positions): same FLOPs and parameter names as
  the real frozen tower, synthetic weights;
* ``HashTokenizer`` -- whitespace words -> ids by CRC32, ``<s>``/``</s>``/``<pad>`` = 0/2/1,
  ``batch_encode_plus(texts, padding='longest', return_tensors='pt') -> BatchEncoding`` (the one
  call the model makes, bdetr.py:164-166).
"""
import zlib

import torch
from transformers import BatchEncoding


class HashTokenizer:
    pad_id, bos_id, eos_id, vocab = 1, 0, 2, 50265

    def batch_encode_plus(self, texts, padding="longest", return_tensors="pt"):
        rows = [[self.bos_id] + [3 + zlib.crc32(w.encode()) % (self.vocab - 3) for w in t.split()]
                + [self.eos_id] for t in texts]
        width = max(len(r) for r in rows)
        ids = torch.full((len(rows), width), self.pad_id, dtype=torch.long)
        att = torch.zeros((len(rows), width), dtype=torch.long)
        for i, r in enumerate(rows):
            ids[i, :len(r)] = torch.tensor(r)
            att[i, :len(r)] = 1
        return BatchEncoding({"input_ids": ids, "attention_mask": att})


def random_roberta_base(seed=0):
    from transformers import RobertaConfig, RobertaModel
    cfg = RobertaConfig(vocab_size=50265, max_position_embeddings=514, type_vocab_size=1,
                        hidden_size=768, num_hidden_layers=12, num_attention_heads=12,
                        intermediate_size=3072, pad_token_id=1, bos_token_id=0, eos_token_id=2)
    with torch.random.fork_rng(devices=[]):
        torch.manual_seed(seed)
        model = RobertaModel(cfg, add_pooling_layer=True)
    return model.eval()


def offline_factory(seed=0):
    return lambda: (HashTokenizer(), random_roberta_base(seed))


def synthetic_utterances(batch, tokens=80, seed=0, max_pad_frac=0.3):
    """``batch`` utterances whose tokenised length (incl. <s>,</s>) is ``tokens`` for the first
    and 70-100 % of it for the rest, i.e. 0-30 % right padding after ``padding='longest'``."""
    import numpy as np
    rng = np.random.default_rng(seed + 15485863)
    out = []
    for b in range(batch):
        n = tokens - 2 if b == 0 else int(tokens - 2 - rng.integers(0, int(tokens * max_pad_frac) + 1))
        out.append(" ".join(f"w{int(x)}" for x in rng.integers(0, 30000, size=max(n, 1))))
    return out
