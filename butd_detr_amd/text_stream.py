"""Text stream of the model (reference: models/bdetr.py:72-83,164-173).

The reference runs a FROZEN RoBERTa-base on every batch (bdetr.py:167): ~13.8 GFLOP per scene at 80 tokens, more
than any single block of the hot path, and its output depends on the utterance only.  Three pieces live here:

* ``UtteranceCache`` -- the language model's hidden states per utterance, resident in HBM (a real training set is
  a few tens of thousands of utterances x ~20 tokens x 768 floats: 2-3 GB of 288).  A batch whose utterances
  are all cached is assembled by ONE row gather and the language model is not run at all.
* ``precision``: "f32" (the reference's arithmetic) or "bf16" (the encoder's linear layers under bf16 autocast, i.e.
  hipBLASLt bf16 GEMMs with fp32 accumulation; LayerNorm / softmax stay fp32; hidden states returned as fp32).
* the tokenizer stays on the host but off the critical path: ``GraphedTrainStep`` tokenises the batch announced
  as ``next_inputs`` while the current one runs (train_step.py).

What makes the cache exact: with right padding RoBERTa's position ids are ``cumsum(ids != pad) * (ids != pad) +
pad_idx`` and padded keys are masked with ``finfo.min``, so (a) an utterance's valid rows do not depend on how far
the batch pads it, and (b) every padded position of an utterance carries the SAME vector (pad token + position
``pad_idx``, attending to the valid keys).  The cache therefore stores ``n + 1`` rows per utterance -- its ``n`` token
rows and its one pad row -- and can rebuild the (B, L, 768) tensor the reference would have computed for ANY batch
composition, padded positions included (the contrastive loss does read them: losses.py:468-469).

Semantics: in ``eval()`` the cache reproduces the encoder to GEMM-reassociation error.  In ``train()`` the reference
leaves RoBERTa's dropout active although the tower is frozen (``model.train()`` reaches it); serving cached rows
removes that noise, so caching while training is opt-in (``cache_in_training=True``) and bench.py's headline does
not use it.
"""
import torch


class UtteranceCache:
    """utterance (str) -> (n + 1, hidden) rows in one HBM store.  ``rows`` is the initial capacity; the store
    doubles up to ``max_rows`` (then new utterances are simply not admitted)."""

    def __init__(self, device, hidden=768, rows=1 << 16, max_rows=1 << 24, dtype=torch.float32,
                 cache_in_training=False):
        self.device, self.hidden, self.max_rows = torch.device(device), hidden, max_rows
        self.store = torch.empty(rows, hidden, dtype=dtype, device=self.device)
        self.index = {}                       # utterance -> (first row, n tokens)
        self.used = 0
        self.hits = self.misses = 0
        self.cache_in_training = cache_in_training

    def __len__(self):
        return len(self.index)

    def bytes(self):
        return self.used * self.hidden * self.store.element_size()

    # ------------------------------------------------------------------ read
    def lookup(self, texts, attention_mask):
        """-> (B, L, hidden) fp32 for the batch as tokenised (``attention_mask`` (B, L), 1 = token), or None unless
        EVERY utterance is cached.  One host-built (B, L) row map + one index_select."""
        spans = [self.index.get(t) for t in texts]
        if any(s is None for s in spans):
            self.misses += 1
            return None
        B, L = attention_mask.shape
        rows = torch.empty(B, L, dtype=torch.int64)
        ar = torch.arange(L)
        for b, (first, n) in enumerate(spans):
            if n > L:
                raise ValueError(f"UtteranceCache: '{texts[b][:40]}...' was cached with {n} tokens, batch has {L}")
            rows[b] = first + torch.clamp(ar, max=n)          # positions >= n -> the utterance's pad row
        self.hits += 1
        out = self.store.index_select(0, rows.reshape(-1).to(self.device, non_blocking=True))
        return out.view(B, L, self.hidden).float()

    # ------------------------------------------------------------------ write
    def _reserve(self, need):
        if self.used + need <= self.store.shape[0]:
            return True
        rows = self.store.shape[0]
        while rows < self.used + need:
            rows *= 2
        if rows > self.max_rows:
            return False
        grown = torch.empty(rows, self.hidden, dtype=self.store.dtype, device=self.device)
        grown[:self.used] = self.store[:self.used]
        self.store = grown
        return True

    def insert(self, texts, hidden, attention_mask):
        """``hidden`` (B, L', hidden) of a batch encoded with AT LEAST ONE padded position per utterance
        (``with_pad_column`` appends one), ``attention_mask`` (B, L') of that encoding."""
        counts = attention_mask.sum(1).tolist()
        for b, (t, n) in enumerate(zip(texts, counts)):
            n = int(n)
            if t in self.index:
                continue
            if n >= hidden.shape[1]:
                raise ValueError("UtteranceCache.insert: the encoding has no padded position for this utterance")
            if not self._reserve(n + 1):
                return
            self.store[self.used:self.used + n + 1] = hidden[b, :n + 1].to(self.store.dtype)
            self.index[t] = (self.used, n)
            self.used += n + 1


def with_pad_column(tokenized, pad_id):
    """The batch with one more (padded) column, so that every utterance owns a padded position."""
    ids, att = tokenized["input_ids"], tokenized["attention_mask"]
    out = {"input_ids": torch.cat([ids, ids.new_full((ids.shape[0], 1), pad_id)], 1),
           "attention_mask": torch.cat([att, att.new_zeros((att.shape[0], 1))], 1)}
    return type(tokenized)(out) if not isinstance(tokenized, dict) else out


def run_language_model(encoder, tokenized, precision="f32"):
    """last_hidden_state of the (frozen) language model, fp32; ``precision`` as in the module docstring."""
    ids = tokenized["input_ids"]
    if precision == "bf16" and ids.is_cuda:
        with torch.autocast("cuda", dtype=torch.bfloat16):
            return encoder(**tokenized).last_hidden_state.float()
    if precision not in ("f32", "bf16"):
        raise ValueError(f"text precision {precision!r}: expected 'f32' or 'bf16'")
    return encoder(**tokenized).last_hidden_state


def encode(encoder, tokenized, texts=None, cache=None, precision="f32", training=False, pad_id=1):
    """The text tower's hidden states for a tokenised batch, through the cache when one is given and allowed."""
    use_cache = cache is not None and texts is not None and (not training or cache.cache_in_training)
    if use_cache and tokenized["input_ids"].is_cuda and torch.cuda.is_current_stream_capturing():
        use_cache = False       # a hit/miss decision is host control flow: it cannot be frozen into a graph
    if not use_cache:
        return run_language_model(encoder, tokenized, precision)
    hit = cache.lookup(texts, tokenized["attention_mask"])
    if hit is not None:
        return hit
    padded = with_pad_column(tokenized, pad_id)
    was_training = encoder.training
    encoder.eval()                            # cached rows are the deterministic (dropout-free) hidden states
    try:
        hidden = run_language_model(encoder, padded, precision)
    finally:
        encoder.train(was_training)
    cache.insert(texts, hidden, padded["attention_mask"])
    return hidden[:, :-1].contiguous()
