"""Functional attention / FFN blocks shared by the encoder and decoder layers.

Two implementations of the same maths sit behind each entry point:

* ``torch`` path  -- plain PyTorch-ROCm ops, the arithmetic of torch/nn/functional.py
  ``multi_head_attention_forward`` (packed in-proj, ``q * sqrt(1/head_dim)``, additive -inf
  key-padding mask, softmax, dropout, bmm, out-proj).  It is the fp32 reference the fused kernels
  are tested against, and what runs for shapes/modes the fused kernels do not cover.
* ``hip`` path    -- hand-written gfx950 kernels (include/butd_attention.h), selected with
  ``set_backend("hip")``.

Block = ``LayerNorm(residual + Dropout(MHA(query, key, value)))`` resp.
``LayerNorm(x + FFN(x))`` -- the post-norm pattern every reference layer uses
(encoder_decoder_layers.py:87-96,149-155,356-404).
"""
import math

import torch
import torch.nn.functional as F

_BACKEND = "torch"


def set_backend(name):
    """'torch' (stock ops) or 'hip' (fused gfx950 kernels; raises if the library lacks them)."""
    global _BACKEND
    if name not in ("torch", "hip"):
        raise ValueError(name)
    if name == "hip":
        from . import fused_attention  # noqa: F401  (fails loudly when the HIP library is missing)
    _BACKEND = name


def get_backend():
    return _BACKEND


_STRICT = False


def set_strict(flag):
    """With the 'hip' backend: raise instead of silently running a set-abstraction level or a feature-propagation
    MLP on the stock-torch ops when the fused path does not cover its configuration (bench.py and the timed-shape
    tests set it, so a fallback cannot hide behind a green run).  Returns the previous setting."""
    global _STRICT
    prev, _STRICT = _STRICT, bool(flag)
    return prev


def fell_back(what, module):
    """Called by a module that takes its stock-torch path while the 'hip' backend is selected."""
    module.last_path = "torch"
    if _STRICT:
        raise RuntimeError(f"{what}: the fused gfx950 path does not cover this configuration and strict mode is on "
                           "(attention_blocks.set_strict)")


def new_step(device):
    """Called once per model forward: advances the fused kernels' dropout step counter."""
    if _BACKEND == "hip" and torch.device(device).type == "cuda":
        from . import fused_attention
        fused_attention.new_step(torch.device(device))


def _mha_torch(attn, query, key, value, key_padding_mask):
    b, lq, d = query.shape
    lk = key.shape[1]
    h, hd = attn.num_heads, attn.head_dim
    w, bias = attn.in_proj_weight, attn.in_proj_bias
    if query is key and key is value:
        q, k, v = F.linear(query, w, bias).chunk(3, dim=-1)
    else:
        q = F.linear(query, w[:d], bias[:d])
        if key is value:
            k, v = F.linear(key, w[d:], bias[d:]).chunk(2, dim=-1)
        else:
            k = F.linear(key, w[d:2 * d], bias[d:2 * d])
            v = F.linear(value, w[2 * d:], bias[2 * d:])
    q = q.reshape(b, lq, h, hd).transpose(1, 2)
    k = k.reshape(b, lk, h, hd).transpose(1, 2)
    v = v.reshape(b, lk, h, hd).transpose(1, 2)
    scores = torch.matmul(q * math.sqrt(1.0 / float(hd)), k.transpose(-1, -2))  # (B,h,Lq,Lk)
    if key_padding_mask is not None:
        scores = scores.masked_fill(key_padding_mask[:, None, None, :], float("-inf"))
    probs = torch.softmax(scores, dim=-1)
    if attn.training and attn.dropout > 0.0:
        probs = F.dropout(probs, p=attn.dropout)
    ctx = torch.matmul(probs, v).transpose(1, 2).reshape(b, lq, d)
    return F.linear(ctx, attn.out_proj.weight, attn.out_proj.bias)


def multi_head_attention(attn, query, key, value, key_padding_mask=None):
    """(B,Lq,d),(B,Lk,d),(B,Lk,d),(B,Lk) bool -> (B,Lq,d)."""
    if _BACKEND == "hip" and query.is_cuda:
        from . import fused_attention
        return fused_attention.multi_head_attention(attn, query, key, value, key_padding_mask)
    return _mha_torch(attn, query, key, value, key_padding_mask)


def attention_block(attn, dropout, norm, *, residual, query, key, value, key_padding_mask=None):
    """LayerNorm(residual + Dropout(MHA(query, key, value)))."""
    if _BACKEND == "hip" and query.is_cuda:
        from . import fused_attention
        return fused_attention.attention_block(attn, dropout, norm, residual, query, key, value,
                                               key_padding_mask)
    out = _mha_torch(attn, query, key, value, key_padding_mask)
    return norm(residual + dropout(out))


def block(attn, dropout, norm, *, x, pos=None, memory=None, key_padding_mask=None, xq_pre=None, next_pos=None, ffn=None,
          hoisted=None):
    """The block in the two shapes the model uses it (every call site of
    encoder_decoder_layers.py:87-122,149-155,179-185,356-404):
        memory is None:  self-attention,  query = key = x (+ pos), value = x
        otherwise:       cross-attention, query = x (+ pos), key = value = memory
    with residual x.  Knowing the structure lets the fused backward return summed gradients.

    Hand-over hints (fused backend; the stock path ignores them and every block computes its own operands):
    ``next_pos`` -- also produce ``y + next_pos`` (from the LayerNorm kernel that writes y), the next block's ``xq_pre``;
    ``ffn``      -- (ffn Sequential, norm): the FFN block that follows (``next_pos`` then belongs to ITS output);
    ``hoisted``  -- (fused_attention.DecoderMemory, layer, name): key / value projections of ``memory`` computed already.
    Returns y when neither next_pos nor ffn is given, else (y, y + next_pos | None)."""
    if _BACKEND == "hip" and x.is_cuda:
        from . import fused_attention
        return fused_attention.block(attn, dropout, norm, x, pos, memory, key_padding_mask, xq_pre, next_pos, ffn, hoisted)
    q = x if pos is None else x + pos
    k, v = (q, x) if memory is None else (memory, memory)
    y = norm(x + dropout(_mha_torch(attn, q, k, v, key_padding_mask)))
    if ffn is not None:
        y = ffn[1](y + ffn[0](y))
    if next_pos is None and ffn is None:
        return y
    return y, None     # (y + next_pos is the fused path's by-product only)


def ffn_block(ffn, norm, x):
    """LayerNorm(x + FFN(x)); ffn = Sequential(Linear, ReLU, Dropout, Linear, Dropout)."""
    if _BACKEND == "hip" and x.is_cuda:
        from . import fused_attention
        return fused_attention.ffn_block(ffn, norm, x)
    return norm(x + ffn(x))
