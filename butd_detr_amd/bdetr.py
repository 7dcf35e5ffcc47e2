"""BeaUTyDETR: the 3D language grounder, MI355X build (models/bdetr.py).

Constructor kwargs are exactly the reference's (bdetr.py:46-52, as passed by
train_dist_mod.py:88-99); ``forward(inputs) -> end_points`` consumes and produces the same dict
keys; sub-module / parameter names equal the reference's so released
checkpoints load with ``strict=True``.

Differences that are deliberate and documented:
* the text tower is built by ``text_encoder_factory`` (default: HF ``roberta-base`` via
  ``from_pretrained`` like bdetr.py:73-75).  Offline boxes inject a random-init RoBERTa-shaped
  encoder and a tokenizer stand-in; parameter names under ``text_encoder.*`` are unaffected.
* ``data/class_embeddings3d.npy`` (bdetr.py:88-91, cwd-relative) is loaded when present; when it is
  absent the embedding keeps its random init and a warning is emitted (a checkpoint overrides it).
"""
import os
import warnings

import numpy as np
import torch
import torch.nn.functional as F
from torch import nn

from . import attention_blocks, graph_audit, switches, text_stream
from .fan_out import fan_out, unstack
from .backbone_module import Pointnet2Backbone
from .encoder_decoder_layers import BiDecoderLayer, BiEncoder, BiEncoderLayer
from .modules import (ClsAgnosticPredictHead, GeneralSamplingModule, PointsObjClsModule,
                      PositionEmbeddingLearned)


def _hf_roberta_factory():
    from transformers import RobertaModel, RobertaTokenizerFast
    return (RobertaTokenizerFast.from_pretrained("roberta-base"),
            RobertaModel.from_pretrained("roberta-base"))


def _align_mlp(d_model):
    return nn.Sequential(nn.Linear(d_model, d_model), nn.ReLU(),
                         nn.Linear(d_model, d_model), nn.ReLU(), nn.Linear(d_model, 64))


class _EmbeddingLookup(torch.autograd.Function):
    """``weight[ids]`` whose backward is one ``index_add_`` (atomic adds, ~10 us) instead of torch's
    sort-based embedding backward (one 95 us single-wave launch at the very end of the backward pass for
    the (B, 132) detected-box class ids)."""

    @staticmethod
    def forward(ctx, weight, ids):
        ctx.save_for_backward(ids)
        ctx.rows = weight.shape[0]
        return weight[ids]

    @staticmethod
    def backward(ctx, grad):
        (ids,) = ctx.saved_tensors
        out = grad.new_zeros((ctx.rows, grad.shape[-1]))
        out.index_add_(0, ids.reshape(-1), grad.reshape(-1, grad.shape[-1]))
        return out, None


def _embedding_lookup(weight, ids):
    if weight.is_cuda and weight.requires_grad and torch.is_grad_enabled():
        return _EmbeddingLookup.apply(weight, ids)
    return torch.nn.functional.embedding(ids, weight)


class BeaUTyDETR(nn.Module):
    """See module docstring.  ``num_encoder_layers`` (default 3) exposes the depth the reference
    hard-codes at bdetr.py:104."""

    def __init__(self, num_class=256, num_obj_class=485, input_feature_dim=3, num_queries=256,
                 num_decoder_layers=6, self_position_embedding="loc_learned",
                 contrastive_align_loss=True, d_model=288, butd=True, pointnet_ckpt=None,
                 self_attend=True, *, text_encoder_factory=None, num_encoder_layers=3,
                 class_embeddings_path="data/class_embeddings3d.npy"):
        super().__init__()
        self.num_queries = num_queries
        self.num_decoder_layers = num_decoder_layers
        self.self_position_embedding = self_position_embedding
        self.contrastive_align_loss = contrastive_align_loss
        self.butd = butd

        # visual encoder
        self.backbone_net = Pointnet2Backbone(input_feature_dim=input_feature_dim, width=1)
        if input_feature_dim == 3 and pointnet_ckpt is not None:
            self.backbone_net.load_state_dict(torch.load(pointnet_ckpt), strict=False)

        # text encoder (frozen)
        self.tokenizer, self.text_encoder = (text_encoder_factory or _hf_roberta_factory)()
        for param in self.text_encoder.parameters():
            param.requires_grad = False
        self.text_projector = nn.Sequential(
            nn.Linear(self.text_encoder.config.hidden_size, d_model),
            nn.LayerNorm(d_model, eps=1e-12), nn.Dropout(0.1))

        # detected-box stream
        if self.butd:
            self.butd_class_embeddings = nn.Embedding(num_obj_class, 768)
            if class_embeddings_path and os.path.exists(class_embeddings_path):
                saved = torch.from_numpy(np.load(class_embeddings_path, allow_pickle=True))
                self.butd_class_embeddings.weight.data.copy_(saved)
            else:
                warnings.warn(f"{class_embeddings_path} not found: butd_class_embeddings keeps "
                              "its random init until a checkpoint is loaded")
            # NB the reference sets .requires_grad on the *module* (bdetr.py:92), which is a no-op:
            # the embedding weight stays trainable and is part of the DDP all-reduce.
            self.butd_class_embeddings.requires_grad = False
            self.class_embeddings = nn.Linear(768, d_model - 128)
            self.box_embeddings = PositionEmbeddingLearned(6, 128)

        # cross-modal encoder
        self.pos_embed = PositionEmbeddingLearned(3, d_model)
        bi_layer = BiEncoderLayer(d_model, dropout=0.1, activation="relu", n_heads=8,
                                  dim_feedforward=256, self_attend_lang=self_attend,
                                  self_attend_vis=self_attend, use_butd_enc_attn=butd)
        self.cross_encoder = BiEncoder(bi_layer, num_encoder_layers)

        # query initialisation
        self.points_obj_cls = PointsObjClsModule(d_model)
        self.gsample_module = GeneralSamplingModule()
        self.decoder_query_proj = nn.Conv1d(d_model, d_model, kernel_size=1)

        def head():
            return ClsAgnosticPredictHead(num_class, 1, num_queries, d_model, objectness=False,
                                          heading=False, compute_sem_scores=True)

        self.proposal_head = head()
        self.decoder = nn.ModuleList(
            BiDecoderLayer(d_model, n_heads=8, dim_feedforward=256, dropout=0.1, activation="relu",
                           self_position_embedding=self_position_embedding, butd=self.butd)
            for _ in range(num_decoder_layers))
        self.prediction_heads = nn.ModuleList(head() for _ in range(num_decoder_layers))

        if contrastive_align_loss:
            self.contrastive_align_projection_image = _align_mlp(d_model)
            self.contrastive_align_projection_text = _align_mlp(d_model)

        self.overlap_text_tower = switches.flag("text_overlap", True)
        # text stream options (text_stream.py): an UtteranceCache in HBM, "f32" | "bf16" language-model arithmetic
        self.text_cache = None
        self.text_precision = "f32"
        self._side_stream = None
        # gradient boundary between the encoder and the decoder (see cut_at_encoder_output)
        self._boundary = None
        self.init_bn_momentum()

    # ------------------------------------------------------------------ backbones
    def tokenize(self, inputs):
        """Host-side part of the forward: tokenizer + H2D copy (bdetr.py:164-166).  Split out so a
        captured hipGraph can replay everything after it (``forward_tokenized``)."""
        return self.tokenizer.batch_encode_plus(
            inputs["text"], padding="longest", return_tensors="pt"
        ).to(inputs["point_clouds"].device)

    @torch.no_grad()
    def encode_text(self, tokenized, texts=None):
        """The FROZEN language model on its own (bdetr.py:80-83 freezes every RoBERTa parameter): its
        output depends on the tokens only, so a training loop may run it for the next batch while the
        current one trains and hand the result in as ``inputs["text_encoder_output"]``.  With ``texts`` (the
        utterances behind ``tokenized``) and ``self.text_cache`` set, cached utterances skip the language model
        (text_stream.py); ``self.text_precision`` "bf16" runs its linear layers on the bf16 matrix cores."""
        return text_stream.encode(self.text_encoder, tokenized, texts, self.text_cache, self.text_precision,
                                  training=self.text_encoder.training,
                                  pad_id=getattr(self.text_encoder.config, "pad_token_id", 1) or 1)

    def text_encoder_is_frozen(self):
        return not any(p.requires_grad for p in self.text_encoder.parameters())

    def _run_text_tower(self, tokenized, end_points, hidden=None, texts=None):
        if hidden is None:
            if self.text_encoder_is_frozen():
                hidden = self.encode_text(tokenized, texts)
            else:
                hidden = self.text_encoder(**tokenized).last_hidden_state
        end_points["text_feats"] = self._project_text(hidden)
        # HF masks are 1 = token; torch attention wants True = padding (bdetr.py:171)
        end_points["text_attention_mask"] = tokenized.attention_mask.ne(1).bool()
        end_points["tokenized"] = tokenized

    def _fused(self, t):
        return t.is_cuda and attention_blocks.get_backend() == "hip"

    def _project_text(self, hidden):
        """``text_projector`` = Linear -> LayerNorm -> Dropout (bdetr.py:76-79); the Linear on the grouped GEMM."""
        if not self._fused(hidden):
            return self.text_projector(hidden)
        from .fused_attention import linear
        lin, norm, drop = self.text_projector
        return drop(norm(linear(lin, hidden)))

    def _run_backbones(self, inputs, tokenized=None):
        """Visual and text towers.  They are independent until the cross-encoder, and the FPS chain of
        the point backbone is a long serial kernel on a handful of CUs -- so on a GPU the text tower is
        issued on a side stream and joins before the encoder (fork/join is capturable in a hipGraph)."""
        if tokenized is None:
            tokenized = self.tokenize(inputs)
        pc = inputs["point_clouds"]
        text_out = {}
        hidden = inputs.get("text_encoder_output")
        if hidden is not None:          # language model already run (prefetched): projector only
            end_points = self.backbone_net(pc, end_points={}, sample_inds=inputs.get("backbone_sample_inds"),
                                           plan=inputs.get("backbone_plan"))
            self._run_text_tower(tokenized, text_out, hidden)
        elif pc.is_cuda and self.overlap_text_tower and self.text_encoder_is_frozen():
            # only the FROZEN language model is forked (no autograd nodes on the side stream: every node's backward
            # runs on the stream of its forward, and tensors that cross streams inside autograd are where captured
            # graphs get their hidden hazards); the trainable projector runs on the main stream after the join
            main = torch.cuda.current_stream(pc.device)
            if self._side_stream is None:
                self._side_stream = graph_audit.own_stream(pc.device, role="model.text", owner=self)
            side = self._side_stream
            side.wait_stream(main)
            with torch.cuda.stream(side):
                hidden = self.encode_text(tokenized, inputs.get("text"))
            end_points = self.backbone_net(pc, end_points={}, sample_inds=inputs.get("backbone_sample_inds"),
                                           plan=inputs.get("backbone_plan"))
            main.wait_stream(side)
            hidden.record_stream(main)
            self._run_text_tower(tokenized, text_out, hidden)
        else:
            end_points = self.backbone_net(pc, end_points={}, sample_inds=inputs.get("backbone_sample_inds"),
                                           plan=inputs.get("backbone_plan"))
            self._run_text_tower(tokenized, text_out, texts=inputs.get("text"))
        end_points["seed_inds"] = end_points["fp2_inds"]
        end_points["seed_xyz"] = end_points["fp2_xyz"]
        end_points["seed_features"] = end_points["fp2_features"]
        end_points.update(text_out)
        return end_points

    def _generate_queries(self, xyz, features, end_points, features_pm=None, forced_seeds=None):
        """bdetr.py:238-249.  ``forced_seeds`` (B, num_queries) int (``inputs["query_seed_inds"]``, not in the reference):
        the seeds to use as queries instead of the top-k by objectness -- a parity hook: it pins the one discrete choice
        of the forward pass, so that a reduced-precision run can be compared query by query with the reference's vectors
        (tests/test_gpu_golden_modules.py)."""
        logits = self.points_obj_cls(features, features_pm=features_pm)
        end_points["seeds_obj_cls_logits"] = logits
        if forced_seeds is not None:
            sample_inds = forced_seeds.to(device=logits.device, dtype=torch.int32).contiguous()
        else:
            sample_inds = torch.topk(torch.sigmoid(logits).squeeze(1), self.num_queries)[1].int()
        xyz, features, sample_inds = self.gsample_module(xyz, features, sample_inds)
        end_points["query_points_xyz"] = xyz
        end_points["query_points_feature"] = features
        end_points["query_points_sample_inds"] = sample_inds
        return end_points

    def _normalized_proj(self, x):
        from .rowwise import l2_normalize       # (F.normalize; one launch each way on the fused backend)
        return l2_normalize(self._proj_mlp(self.contrastive_align_projection_image, x))

    @staticmethod
    def _proj_mlp(seq, x):
        """The Linear-ReLU-Linear-ReLU-Linear projections on the grouped GEMM (fused gate / bias / ReLU epilogues)
        when the fused backend is active; the stock modules otherwise."""
        if x.is_cuda and attention_blocks.get_backend() == "hip":
            from .fused_attention import linear_relu_chain
            return linear_relu_chain(seq, x)
        return seq(x)

    # ------------------------------------------------------------------ forward
    def forward(self, inputs):
        """inputs: point_clouds (B,N,3+C), text list[str]; with butd also det_boxes (B,132,6),
        det_bbox_label_mask (B,132) bool, det_class_ids (B,132) i64.  Returns ``end_points``."""
        return self.forward_tokenized(inputs, None)

    def forward_tokenized(self, inputs, tokenized):
        """Device-side forward given an already tokenised (and device-resident) utterance batch;
        ``tokenized=None`` tokenises here (the reference behaviour)."""
        attention_blocks.new_step(inputs["point_clouds"].device)
        end_points = self._run_backbones(inputs, tokenized)
        points_xyz = end_points["fp2_xyz"]                       # (B, V, 3)
        points_features = end_points["fp2_features"]             # (B, d, V)
        text_feats = end_points["text_feats"]                    # (B, L, d)
        text_padding_mask = end_points["text_attention_mask"]    # (B, L)

        detected_mask = detected_feats = None
        if self.butd:
            detected_mask = ~inputs["det_bbox_label_mask"]
            class_feats = _embedding_lookup(self.butd_class_embeddings.weight, inputs["det_class_ids"])
            if self._fused(class_feats):
                from .fused_attention import linear
                class_feats = linear(self.class_embeddings, class_feats)
            else:
                class_feats = self.class_embeddings(class_feats)
            detected_feats = torch.cat(
                [self.box_embeddings(inputs["det_boxes"]), class_feats.transpose(1, 2)], 1
            ).transpose(1, 2).contiguous()                       # (B, D, d)

        self._stage_hook(inputs, "encoder")
        vis, text_feats = self.cross_encoder(
            vis_feats=points_features.transpose(1, 2).contiguous(),
            pos_feats=self.pos_embed(points_xyz).transpose(1, 2).contiguous(),
            padding_mask=torch.zeros(points_xyz.shape[:2], dtype=torch.bool,
                                     device=points_xyz.device),
            text_feats=text_feats, text_padding_mask=text_padding_mask, end_points=end_points,
            detected_feats=detected_feats, detected_mask=detected_mask)
        vis, text_feats, detected_feats = self._cut(vis, text_feats, detected_feats)
        points_features = vis.transpose(1, 2).contiguous()       # (B, d, V)
        end_points["text_memory"] = text_feats
        end_points["seed_features"] = points_features
        if self.contrastive_align_loss:
            from .rowwise import l2_normalize
            end_points["proj_tokens"] = l2_normalize(self._proj_mlp(self.contrastive_align_projection_text, text_feats))

        end_points = self._generate_queries(points_xyz, points_features, end_points, features_pm=vis,
                                            forced_seeds=inputs.get("query_seed_inds"))
        cluster_feature = end_points["query_points_feature"]     # (B, d, Q)
        cluster_xyz = end_points["query_points_xyz"]             # (B, Q, 3)
        if (self._fused(cluster_feature) and self.decoder_query_proj.in_channels % 4 == 0
                and self.decoder_query_proj.out_channels % 4 == 0):
            from .fused_attention import conv1x1      # (the 1x1 convolution on position-major rows: bdetr.py:296)
            query = conv1x1(self.decoder_query_proj, cluster_feature.transpose(1, 2))
        else:
            query = self.decoder_query_proj(cluster_feature).transpose(1, 2).contiguous()
        # contrastive projections of the proposal / per-layer queries (bdetr.py:263-268,300-305): the same
        # MLP on seven tensors that nothing downstream of the model reads before the loss -- collected
        # here and projected as ONE stacked batch after the decoder (row-wise op: identical values)
        proj_inputs = [("proposal_", query)] if self.contrastive_align_loss else []

        center, size = self._run_head(self.proposal_head, cluster_feature, cluster_xyz, end_points, "proposal_")
        base_xyz, base_size = center.detach(), size.detach()   # (the reference clones: bdetr.py:275-276; the cat /
            # position embedding below copy them anyway and nothing writes the head outputs in place)

        # the encoder outputs feed all decoder layers: one gradient sum per stream (fan_out.py)
        self._stage_hook(inputs, "decoder")
        n_dec = len(self.decoder)
        # the three memories' key / value projections of ALL layers in a few grouped launches before the decoder, their
        # input / weight gradients in one node after its backward (fused_attention.DecoderMemory; fused backend only)
        memory_kv = None
        if self._fused(vis):
            from .fused_attention import decoder_memory
            memory_kv = decoder_memory(list(self.decoder), [("text", "cross_l", text_feats),
                                                              ("boxes", "cross_d", detected_feats if self.butd else None),
                                                              ("seeds", "cross_v", vis)])
        if memory_kv is None:
            vis_l, text_l = fan_out(vis, n_dec), fan_out(text_feats, n_dec)
            det_l = fan_out(detected_feats if self.butd else None, n_dec)
        else:       # (no per-layer gradient of the memories arrives: the hoisted node returns their sums)
            vis_l, text_l = (vis.detach(),) * n_dec, (text_feats.detach(),) * n_dec
            det_l = ((detected_feats.detach() if self.butd else None),) * n_dec
        for i, (layer, head) in enumerate(zip(self.decoder, self.prediction_heads)):
            prefix = "last_" if i == self.num_decoder_layers - 1 else f"{i}head_"
            if self.self_position_embedding == "none":
                query_pos = None
            elif self.self_position_embedding == "xyz_learned":
                query_pos = base_xyz
            elif self.self_position_embedding == "loc_learned":
                query_pos = torch.cat([base_xyz, base_size], -1)
            else:
                raise NotImplementedError
            query = layer(query, vis_l[i], text_l[i], query_pos, None, text_padding_mask,
                          detected_feats=det_l[i],
                          detected_mask=detected_mask if self.butd else None,
                          **({} if memory_kv is None else {"memory_kv": (memory_kv, i)}))
            # the layer output feeds the next layer, its head and the contrastive projection
            query, q_head, q_proj = fan_out(query, 3)
            if self.contrastive_align_loss:
                proj_inputs.append((prefix, q_proj))
            center, size = self._run_head(head, q_head.transpose(1, 2), cluster_xyz, end_points, prefix, features_pm=q_head)
            base_xyz, base_size = center.detach(), size.detach()   # (the reference clones: bdetr.py:275-276; the cat /
            # position embedding below copy them anyway and nothing writes the head outputs in place)
        if memory_kv is not None:
            memory_kv.token = None      # (the blocks hold the autograd edge; this breaks the token -> node -> object cycle)
        if proj_inputs:
            proj = unstack(self._normalized_proj(torch.stack([q for _, q in proj_inputs])))
            for i, (prefix, _) in enumerate(proj_inputs):
                end_points[f"{prefix}proj_queries"] = proj[i]
        return end_points

    @staticmethod
    def _stage_hook(inputs, stage):
        """Optional callbacks of the caller at stage boundaries of the forward pass (``inputs["_stage_hooks"] =
        {"encoder": fn, "decoder": fn}``): the captured training step uses them to fork its prefetch branches (the next
        batch's language model / sampling chain) where the main queue leaves the chip idle instead of at the start of
        the step (train_step.GraphedTrainStep)."""
        hooks = inputs.get("_stage_hooks")
        if hooks and stage in hooks:
            hooks[stage]()

    def _run_head(self, head, features, cluster_xyz, end_points, prefix, features_pm=None):
        """One prediction head (bdetr.py:306-312).  (Round 4 measured the heads on a forked stream -- their backward depends
        on the loss alone -- at +0.18 ms per step: two queues of small launches share the chip no better than one;
        profiles/r04_side_branches.txt.  Removed in round 5.)"""
        return head(features, base_xyz=cluster_xyz, end_points=end_points, prefix=prefix, features_pm=features_pm)

    # parameters whose gradients are complete only once backward has passed the encoder: everything else
    # (decoder, heads, query generation, contrastive projections) is done when backward reaches the three
    # tensors the decoder reads from the encoder side -- the first bucket of an overlapped gradient exchange
    # (DistributedDataParallel's buckets do the same in reverse registration order, main_utils.py:310-313)
    pre_boundary_prefixes = ("backbone_net.", "text_encoder.", "text_projector.", "pos_embed.", "box_embeddings.",
                             "class_embeddings.", "butd_class_embeddings.", "cross_encoder.")

    def cut_at_encoder_output(self, enable=True):
        """Two-stage backward: with the cut enabled a forward hands the decoder DETACHED copies of the encoder
        outputs (visual features, text features, box features) and records (outputs, copies) in
        ``self._boundary``; ``loss.backward()`` then stops at the copies (stage 1: decoder-side gradients
        final), and ``torch.autograd.backward(outputs, [c.grad for c in copies])`` runs the rest (stage 2)."""
        self._boundary = [] if enable else None

    def _cut(self, *tensors):
        if self._boundary is None or not torch.is_grad_enabled():
            return tensors
        outs = [t for t in tensors if t is not None and t.requires_grad]
        copies = {id(t): t.detach().requires_grad_(True) for t in outs}
        self._boundary.append((outs, [copies[id(t)] for t in outs]))
        return tuple(copies.get(id(t), t) if t is not None else None for t in tensors)

    def init_bn_momentum(self):
        for m in self.modules():
            if isinstance(m, (nn.BatchNorm2d, nn.BatchNorm1d)):
                m.momentum = 0.1
