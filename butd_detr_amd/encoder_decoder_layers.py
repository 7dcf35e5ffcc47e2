"""Cross-modal encoder / decoder layers of BUTD-DETR (models/encoder_decoder_layers.py).

Same classes, constructor arguments, forward signatures and parameter names as the reference --
``PositionEmbeddingLearned`` (:19-34), ``CrossAttentionLayer`` (:37-124),
``TransformerEncoderLayerNoFFN`` (:127-156), ``PosTransformerEncoderLayerNoFFN`` (:159-186),
``BiEncoderLayer`` (:189-255), ``BiEncoder`` (:258-284), ``BiDecoderLayer`` (:287-406) -- but all
tensors stay batch-first (B, L, d): the reference's seq-first transposes only exist to feed
``nn.MultiheadAttention(batch_first=False)`` and are arithmetic no-ops.

Every attention / FFN block is routed through ``attention_blocks`` (fused gfx950 kernels when
enabled, plain torch ops otherwise -- same maths as torch/nn/functional.py multi_head_attention_forward:
packed in-proj, q * sqrt(1/head_dim), additive -inf key-padding mask, softmax, dropout, out-proj).
"""
from copy import deepcopy

import os

import torch
from torch import nn

from . import attention_blocks as ab
from . import graph_audit, switches
from .fan_out import fan_out


def _get_clones(module, n):
    return nn.ModuleList([deepcopy(module) for _ in range(n)])


class PositionEmbeddingLearned(nn.Module):
    """Conv1d(k->F) + BN1d + ReLU + Conv1d(F->F): xyz (B,N,k) -> (B,F,N)."""

    def __init__(self, input_channel, num_pos_feats=288):
        super().__init__()
        self.position_embedding_head = nn.Sequential(
            nn.Conv1d(input_channel, num_pos_feats, kernel_size=1),
            nn.BatchNorm1d(num_pos_feats),
            nn.ReLU(inplace=True),
            nn.Conv1d(num_pos_feats, num_pos_feats, kernel_size=1))

    def forward(self, xyz):
        if ab.get_backend() == "hip" and xyz.is_cuda:
            from .fused_mlp import mlp_chains
            head = self.position_embedding_head
            b, n, k = xyz.shape
            out = mlp_chains(xyz.reshape(b * n, k), [([(head[0], head[1])], head[3], 0.0)],
                             self.training)[0]
            return out.view(b, n, -1).transpose(1, 2)     # (B,F,N) view of the position-major result
        return self.position_embedding_head(xyz.transpose(1, 2).contiguous())


class MultiheadAttention(nn.Module):
    """Parameter container with ``nn.MultiheadAttention``'s names and init (``in_proj_weight``
    (3d,d) xavier-uniform, ``in_proj_bias`` zeros, ``out_proj.{weight,bias}``), batch-first maths.

    forward(query (B,Lq,d), key (B,Lk,d), value (B,Lk,d), key_padding_mask (B,Lk) bool | None)
        -> (B,Lq,d)  [the reference discards the averaged attention weights with ``[0]``]
    """

    def __init__(self, embed_dim, num_heads, dropout=0.0):
        super().__init__()
        assert embed_dim % num_heads == 0
        self.embed_dim, self.num_heads, self.dropout = embed_dim, num_heads, dropout
        self.head_dim = embed_dim // num_heads
        self.in_proj_weight = nn.Parameter(torch.empty(3 * embed_dim, embed_dim))
        self.in_proj_bias = nn.Parameter(torch.empty(3 * embed_dim))
        self.out_proj = nn.Linear(embed_dim, embed_dim)
        nn.init.xavier_uniform_(self.in_proj_weight)
        nn.init.constant_(self.in_proj_bias, 0.0)
        nn.init.constant_(self.out_proj.bias, 0.0)

    def forward(self, query, key, value, key_padding_mask=None):
        return ab.multi_head_attention(self, query, key, value, key_padding_mask)


def _ffn(d_model, dim_feedforward, dropout):
    return nn.Sequential(
        nn.Linear(d_model, dim_feedforward), nn.ReLU(), nn.Dropout(dropout),
        nn.Linear(dim_feedforward, d_model), nn.Dropout(dropout))


class CrossAttentionLayer(nn.Module):
    """text<-vision, vision<-text, [vision<-boxes], FFNs; post-LayerNorm everywhere."""

    def __init__(self, d_model=256, dropout=0.1, n_heads=8, dim_feedforward=256,
                 use_butd_enc_attn=False):
        super().__init__()
        self.use_butd_enc_attn = use_butd_enc_attn
        self.cross_lv = MultiheadAttention(d_model, n_heads, dropout=dropout)
        self.dropout_lv = nn.Dropout(dropout)
        self.norm_lv = nn.LayerNorm(d_model)
        self.ffn_lv = _ffn(d_model, dim_feedforward, dropout)
        self.norm_lv2 = nn.LayerNorm(d_model)

        self.cross_vl = deepcopy(self.cross_lv)
        self.dropout_vl = nn.Dropout(dropout)
        self.norm_vl = nn.LayerNorm(d_model)
        self.ffn_vl = deepcopy(self.ffn_lv)
        self.norm_vl2 = nn.LayerNorm(d_model)

        if use_butd_enc_attn:
            self.cross_d = MultiheadAttention(d_model, n_heads, dropout=dropout)
            self.dropout_d = nn.Dropout(dropout)
            self.norm_d = nn.LayerNorm(d_model)

    def forward(self, vis_feats, vis_key_padding_mask, text_feats, text_key_padding_mask,
                pos_feats, detected_feats=None, detected_mask=None):
        """vis/pos (B,V,d), text (B,L,d), boxes (B,D,d); masks True = padding."""
        # the two branches only READ each other's layer input (:83,:101-102), so they are independent
        text_out = self.language_branch(text_feats, vis_feats, vis_key_padding_mask)
        vis_out = self.vision_branch(vis_feats, text_feats, text_key_padding_mask, pos_feats,
                                     detected_feats, detected_mask)
        return vis_out, text_out

    def language_branch(self, text_feats, vis_feats, vis_key_padding_mask):
        """language attends to vision, then its FFN"""
        return ab.block(self.cross_lv, self.dropout_lv, self.norm_lv, x=text_feats, memory=vis_feats,
                        key_padding_mask=vis_key_padding_mask, ffn=(self.ffn_lv, self.norm_lv2))[0]

    def vision_branch(self, vis_feats, text_in, text_key_padding_mask, pos_feats,
                      detected_feats=None, detected_mask=None, xq_pre=None, next_pos=None):
        """vision attends to language (keys/values = the layer INPUT text), [to the boxes], FFN.
        ``next_pos``: -> (out, out + next_pos): the `src + pos` of the next layer's self-attention, from the FFN's
        LayerNorm kernel (values only; None on the stock path)."""
        # positional features only on the query (:79-80)
        boxes = detected_feats is not None and self.use_butd_enc_attn
        if boxes:
            vis_feats = ab.block(self.cross_vl, self.dropout_vl, self.norm_vl, x=vis_feats, pos=pos_feats,
                                 memory=text_in, key_padding_mask=text_key_padding_mask, xq_pre=xq_pre)
            vis_feats, vis_pos = ab.block(self.cross_d, self.dropout_d, self.norm_d, x=vis_feats,
                                          memory=detected_feats, key_padding_mask=detected_mask,
                                          ffn=(self.ffn_vl, self.norm_vl2), next_pos=next_pos)
        else:
            vis_feats, vis_pos = ab.block(self.cross_vl, self.dropout_vl, self.norm_vl, x=vis_feats, pos=pos_feats,
                                          memory=text_in, key_padding_mask=text_key_padding_mask, xq_pre=xq_pre,
                                          ffn=(self.ffn_vl, self.norm_vl2), next_pos=next_pos)
        return vis_feats if next_pos is None else (vis_feats, vis_pos)


class TransformerEncoderLayerNoFFN(nn.Module):
    """Self-attention + residual + LayerNorm, no FFN.  src (B,S,d)."""

    def __init__(self, d_model, nhead, dropout):
        super().__init__()
        self.self_attn = MultiheadAttention(d_model, nhead, dropout=dropout)
        self.norm1 = nn.LayerNorm(d_model)
        self.dropout1 = nn.Dropout(dropout)

    def forward(self, src, src_mask=None, src_key_padding_mask=None):
        assert src_mask is None, "attn_mask is never used on this path"
        return ab.block(self.self_attn, self.dropout1, self.norm1, x=src, key_padding_mask=src_key_padding_mask)


class PosTransformerEncoderLayerNoFFN(TransformerEncoderLayerNoFFN):
    """Same, with the positional embedding added to query and key (not value)."""

    def forward(self, src, pos, src_mask=None, src_key_padding_mask=None, xq_pre=None, with_pos=False):
        """``with_pos``: -> (out, out + pos | None): the sum the following cross-attention takes as its query input, from
        the LayerNorm kernel that writes out (values only; None on the stock path).
        ``xq_pre``: src + pos as an earlier kernel already wrote it (values only)."""
        assert src_mask is None, "attn_mask is never used on this path"
        return ab.block(self.self_attn, self.dropout1, self.norm1, x=src, pos=pos,
                        key_padding_mask=src_key_padding_mask, xq_pre=xq_pre, next_pos=pos if with_pos else None)


_LANGUAGE_STREAMS = {}
_ENCODER_FORK = [switches.flag("encoder_fork", True)]


def _language_stream(device):
    st = _LANGUAGE_STREAMS.get(device)
    if st is None:
        st = _LANGUAGE_STREAMS[device] = graph_audit.own_stream(device, role="encoder.language")
    return st


def _fork_language(t):
    """Fused gfx950 path on a GPU only (BUTD_AB=encoder_fork=0 switches it off: A/B hook, worth 0.65 ms per step)."""
    return t.is_cuda and ab.get_backend() == "hip" and _ENCODER_FORK[0]


class BiEncoderLayer(nn.Module):
    """visual self-attn -> language self-attn -> CrossAttentionLayer."""

    def __init__(self, d_model=256, dropout=0.1, activation="relu", n_heads=8,
                 dim_feedforward=256, self_attend_lang=True, self_attend_vis=True,
                 use_butd_enc_attn=False):
        super().__init__()
        self.self_attention_lang = (TransformerEncoderLayerNoFFN(d_model, n_heads, dropout)
                                    if self_attend_lang else None)
        self.self_attention_visual = (PosTransformerEncoderLayerNoFFN(d_model, n_heads, dropout)
                                      if self_attend_vis else None)
        self.cross_layer = CrossAttentionLayer(d_model, dropout, n_heads, dim_feedforward,
                                               use_butd_enc_attn)

    def forward(self, vis_feats, pos_feats, padding_mask, text_feats, text_padding_mask,
                end_points={}, detected_feats=None, detected_mask=None, vis_xq_pre=None, next_pos=None):
        """The reference's signature (:212-255) plus two optional hints of the layer stack: ``pos_feats`` may be a pair
        (alias for the self-attention, alias for the cross-attention: one gradient sum for all layers' aliases,
        fan_out.py); ``vis_xq_pre`` = vis_feats + pos as the previous layer's last kernel wrote it; ``next_pos`` ->
        a third result, this layer's vis output + next_pos, for the next layer."""
        pos_self, pos_cross = pos_feats if isinstance(pos_feats, (tuple, list)) else (pos_feats, pos_feats)
        fork = _fork_language(vis_feats)
        if fork:
            main = torch.cuda.current_stream(vis_feats.device)
            side = _language_stream(vis_feats.device)
        else:
            main = side = None
        from contextlib import nullcontext
        on_side = (lambda: torch.cuda.stream(side)) if fork else nullcontext
        # The language side of the layer (self-attention over <= 80 tokens, language <- vision cross-attention, its
        # FFN: 640-row kernels that leave the chip idle) runs on a forked stream next to the vision side, which is
        # several times longer.  The two sides only read each other's self-attention outputs
        # (encoder_decoder_layers.py:83,101-102), so there are two hand-over points; fork / join are captured as
        # parallel branches of the hipGraph, autograd runs each node's backward on the stream of its forward.
        cross = self.cross_layer
        if fork:
            side.wait_stream(main)
            text_feats.record_stream(side)
        with on_side():
            text_self = text_feats
            if self.self_attention_lang is not None:
                text_self = self.self_attention_lang(text_feats, src_key_padding_mask=text_padding_mask)
        vis_self, vis_self_pos = vis_feats, None
        if self.self_attention_visual is not None:
            vis_self, vis_self_pos = self.self_attention_visual(vis_feats, pos_self, src_key_padding_mask=padding_mask,
                                                                xq_pre=vis_xq_pre, with_pos=True)
        if fork:
            side.wait_stream(main)                  # language <- vision reads vis_self
            main.wait_stream(side)                  # vision <- language reads text_self
            vis_self.record_stream(side)
            text_self.record_stream(main)
        with on_side():
            text_out = cross.language_branch(text_self, vis_self, padding_mask)
        vis_out = cross.vision_branch(vis_self, text_self, text_padding_mask, pos_cross, detected_feats,
                                      detected_mask, xq_pre=vis_self_pos, next_pos=next_pos)
        if fork:
            main.wait_stream(side)                  # join
            text_out.record_stream(main)
        if next_pos is not None:
            return vis_out[0], text_out, vis_out[1]
        return vis_out, text_out


class BiEncoder(nn.Module):
    def __init__(self, bi_layer, num_layers):
        super().__init__()
        self.layers = _get_clones(bi_layer, num_layers)
        self.num_layers = num_layers

    def forward(self, vis_feats, pos_feats, padding_mask, text_feats, text_padding_mask,
                end_points={}, detected_feats=None, detected_mask=None):
        # the position embedding and the box stream feed every layer: one gradient sum each (fan_out.py)
        # (two consumers per layer -- the self-attention and the vision <- language cross-attention: 2 n aliases, ONE sum)
        # (the one-launch sum takes up to 8 sources: deeper stacks keep one alias per layer)
        handoff = True          # (`src + pos` handed from layer to layer; A/B of round 4: neutral, 5 launches fewer)
        two = handoff and 2 * self.num_layers <= 8
        pos_l = fan_out(pos_feats, 2 * self.num_layers if two else self.num_layers)
        pos_of = (lambda i: (pos_l[2 * i], pos_l[2 * i + 1])) if two else (lambda i: pos_l[i])
        det_l = fan_out(detected_feats, self.num_layers)
        xq_pre = None
        for i, layer in enumerate(self.layers):
            # every layer but the last also writes `its output + pos` from its last LayerNorm kernel: the next layer's
            # query / key input (values only: the fused path's by-product, None on the stock path)
            nxt = pos_feats if (handoff and i + 1 < self.num_layers and pos_feats is not None) else None
            out = layer(vis_feats, pos_of(i), padding_mask, text_feats,
                        text_padding_mask, end_points, detected_feats=det_l[i], detected_mask=detected_mask,
                        vis_xq_pre=xq_pre, next_pos=nxt)
            vis_feats, text_feats = out[0], out[1]
            xq_pre = out[2] if len(out) > 2 else None
            if "lv_attention" in end_points:
                end_points["lv_attention%d" % i] = end_points["lv_attention"]
        return vis_feats, text_feats


class BiDecoderLayer(nn.Module):
    """query self-attn -> cross to language -> [cross to boxes] -> cross to vision -> FFN."""

    def __init__(self, d_model, n_heads, dim_feedforward=2048, dropout=0.1, activation="relu",
                 self_position_embedding="loc_learned", butd=False):
        super().__init__()
        self.self_attn = MultiheadAttention(d_model, n_heads, dropout=dropout)
        self.norm1 = nn.LayerNorm(d_model)
        self.dropout1 = nn.Dropout(dropout)

        self.cross_l = MultiheadAttention(d_model, n_heads, dropout=dropout)
        self.dropout_l = nn.Dropout(dropout)
        self.norm_l = nn.LayerNorm(d_model)

        if butd:
            self.cross_d = deepcopy(self.cross_l)
            self.dropout_d = nn.Dropout(dropout)
            self.norm_d = nn.LayerNorm(d_model)

        self.cross_v = deepcopy(self.cross_l)
        self.dropout_v = nn.Dropout(dropout)
        self.norm_v = nn.LayerNorm(d_model)

        self.ffn = _ffn(d_model, dim_feedforward, dropout)
        self.norm2 = nn.LayerNorm(d_model)

        if self_position_embedding == "xyz_learned":
            self.self_posembed = PositionEmbeddingLearned(3, d_model)
        elif self_position_embedding == "loc_learned":
            self.self_posembed = PositionEmbeddingLearned(6, d_model)
        else:
            self.self_posembed = None

    def forward(self, query, vis_feats, lang_feats, query_pos, padding_mask,
                text_key_padding_mask, detected_feats=None, detected_mask=None, memory_kv=None):
        """query (B,Q,d), vis (B,V,d), lang (B,L,d), query_pos (B,Q,3|6) -> (B,Q,d).
        ``memory_kv`` (not in the reference): (fused_attention.DecoderMemory, layer index) -- the key / value projections of
        the three memories were computed for all layers before the decoder (bdetr.py); None: each block projects its own."""
        if self.self_posembed is not None:
            query_pos = self.self_posembed(query_pos).transpose(1, 2).contiguous()
        else:
            query_pos = None       # the reference adds zeros_like(query) (:364-365)

        # the four blocks' position gradients arrive together and are summed in one pass (fan_out.py)
        pos_s, pos_l, pos_d, pos_v = fan_out(query_pos, 4)
        # ... and every block's LayerNorm kernel also writes `its output + query_pos`: the next block's query input
        nxt = query_pos

        def blk(*args, **kw):
            out = ab.block(*args, next_pos=nxt, **kw)
            return out if isinstance(out, tuple) else (out, None)

        boxes = detected_feats is not None
        hoisted = (lambda name: None) if memory_kv is None else (lambda name: (memory_kv[0], memory_kv[1], name))
        query, qp = blk(self.self_attn, self.dropout1, self.norm1, x=query, pos=pos_s, key_padding_mask=padding_mask)
        query, qp = blk(self.cross_l, self.dropout_l, self.norm_l, x=query, pos=pos_l, xq_pre=qp, memory=lang_feats,
                        key_padding_mask=text_key_padding_mask, hoisted=hoisted("text"))
        if boxes:
            query, qp = blk(self.cross_d, self.dropout_d, self.norm_d, x=query, pos=pos_d, xq_pre=qp,
                            memory=detected_feats, key_padding_mask=detected_mask, hoisted=hoisted("boxes"))
        query = ab.block(self.cross_v, self.dropout_v, self.norm_v, x=query, pos=pos_v, xq_pre=qp,
                         memory=vis_feats, key_padding_mask=None, ffn=(self.ffn, self.norm2), hoisted=hoisted("seeds"))[0]
        return query.contiguous()
