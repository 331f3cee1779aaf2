"""Audio-seeded masked transformer decoder, mirroring
models/modeling/transformer_decoder/transformer_decoder.py: same class names, constructor arguments,
parameter names (so reference checkpoints load 1:1) and the same output dict.

MI355X notes: activations are batch-first [BT, Q, C] / token-major [BT, hw, C] (the reference is sequence-first);
position encodings are cached; the boolean attention mask is kept as ONE [BT, Q, hw] tensor and broadcast over
the 8 heads instead of being materialised 8x (transformer_decoder.py:504) and feeds the head's own attention kernels
(csrc/attention.hip: exact-fp32 MFMA, transposed score tiles); the "fully blocked row" reset
(:458, a nonzero()+index_put => host sync in the reference) is a sync-free logical op; mask logits come from a
token-major pixel embedding so the contraction is a K-contiguous NT GEMM.
"""
import logging

import torch
from torch import nn
from torch.nn import functional as F

from ..registry import TRANSFORMER_DECODER_REGISTRY
from ..ops.linear import Linear, ffn, in_proj, in_proj_q, linear, memory_kv
from .layers import MLP, position_embedding_sine


from ..ops.colsum import add_channel_vector


def _deferred_layer_norm(dim):
    """per-layer post-norm: applied once per forward, so its parameter gradients may join the grouped launch"""
    from ..ops.layernorm import LayerNorm
    ln = LayerNorm(dim)
    ln.defer_dw = True
    return ln


class _MHAParams(nn.Module):
    """Parameter container with nn.MultiheadAttention's names (in_proj_weight/in_proj_bias/out_proj.*)."""

    def __init__(self, d_model, nhead):
        super().__init__()
        self.embed_dim, self.num_heads = d_model, nhead
        self.in_proj_weight = nn.Parameter(torch.empty(3 * d_model, d_model))
        self.in_proj_bias = nn.Parameter(torch.zeros(3 * d_model))
        self.out_proj = Linear(d_model, d_model)
        self.out_proj.defer_dw = True
        nn.init.xavier_uniform_(self.in_proj_weight)
        nn.init.constant_(self.out_proj.bias, 0.0)

    def forward(self, query, key, value, blocked=None, kv=None):
        """batch-first: query [B,Lq,E], key/value [B,Lk,E]; blocked: an ops.masklogit.PackedMask / uint8 [B,Lq,pitch] (1 = masked
        out, rows padded to a multiple of 4 bytes) or None.  kv: this layer's (k, v) row views [B*Lk, E] out of
        ops.linear.memory_kv (the projections of all layers that share the memory, computed together) instead of key / value."""
        from ..ops.attention import attention
        E, H = self.embed_dim, self.num_heads
        B, Lq, _ = query.shape
        # (these weights are used once per forward: their dW GEMMs may be deferred into the grouped launch, ops/linear.py)
        if kv is not None:
            q = in_proj_q(query, self.in_proj_weight, self.in_proj_bias, defer=True)
            o = attention(q.reshape(B * Lq, E), kv[0], kv[1], blocked, B, H)  # csrc/attention.hip
            return self.out_proj(o.view(B, Lq, E))
        Lk = key.shape[1]
        q, k, v = in_proj(query, key, value, self.in_proj_weight, self.in_proj_bias, same_qk=query is key, defer=True)
        o = attention(q.reshape(B * Lq, E), k.reshape(B * Lk, E), v.reshape(B * Lk, E), blocked, B, H)
        return self.out_proj(o.view(B, Lq, E))


class SelfAttentionLayer(nn.Module):
    def __init__(self, d_model, nhead, dropout=0.0, activation="relu", normalize_before=False):
        super().__init__()
        assert not normalize_before and dropout == 0.0, "shipped configs: PRE_NORM False, dropout 0"
        self.self_attn = _MHAParams(d_model, nhead)
        self.norm = _deferred_layer_norm(d_model)
        self._reset_parameters()

    def _reset_parameters(self):
        for p in self.parameters():
            if p.dim() > 1:
                nn.init.xavier_uniform_(p)

    def forward(self, tgt, query_pos, fan=None):
        """fan = (residual, value, query-key = tgt + query_pos) handles on the cross-attention layer's output (see
        ops/layernorm.py: aliases whose gradients are summed inside the LayerNorm backward kernel); returns
        (FFN input, residual) handles on this layer's output."""
        t_res, t_val, qk = fan if fan is not None else (tgt, tgt, tgt + query_pos)
        return self.norm(t_res, self.self_attn(qk, qk, t_val), fanout=2)  # LN(tgt + attn), transformer_decoder.py:50-58


class CrossAttentionLayer(nn.Module):
    def __init__(self, d_model, nhead, dropout=0.0, activation="relu", normalize_before=False):
        super().__init__()
        assert not normalize_before and dropout == 0.0
        self.multihead_attn = _MHAParams(d_model, nhead)
        self.norm = _deferred_layer_norm(d_model)
        self._reset_parameters()

    def _reset_parameters(self):
        for p in self.parameters():
            if p.dim() > 1:
                nn.init.xavier_uniform_(p)

    def forward(self, tgt, memory, blocked, pos, query_pos, memory_k=None, fan=None, kv=None):
        """memory_k: `memory + pos`, computed once per level by the caller (three layers share a level's memory); kv: this
        layer's projected (k, v) when the caller projected the memory for all layers of the level (then memory / pos are unused);
        fan = (residual, query = tgt + query_pos) handles on the previous FFN layer's output; returns
        (residual, value, query-key = out + query_pos) handles on this layer's output for the self-attention layer."""
        if memory_k is None and kv is None:
            memory_k = memory + pos
        t_res, t_q = fan if fan is not None else (tgt, tgt + query_pos)
        tgt2 = self.multihead_attn(t_q, memory_k, memory, blocked, kv=kv)  # :99-118
        return self.norm(t_res, tgt2, fanout=2, pos=query_pos)  # LN(tgt + tgt2) in one pass (csrc/layernorm.hip)


class FFNLayer(nn.Module):
    def __init__(self, d_model, dim_feedforward=2048, dropout=0.0, activation="relu", normalize_before=False):
        super().__init__()
        assert not normalize_before and dropout == 0.0
        self.linear1 = Linear(d_model, dim_feedforward)
        self.linear2 = Linear(dim_feedforward, d_model)
        self.linear2.defer_dw = True
        self.norm = _deferred_layer_norm(d_model)
        self._reset_parameters()

    def _reset_parameters(self):
        for p in self.parameters():
            if p.dim() > 1:
                nn.init.xavier_uniform_(p)

    def forward(self, tgt, fan=None, query_pos=None, last=False):
        """fan = (FFN input, residual) handles on the self-attention layer's output; returns (prediction-head input, residual,
        query = out + query_pos) handles for the prediction head and the next cross-attention layer (last: only the first)."""
        t_in, t_res = fan if fan is not None else (tgt, tgt)
        y = ffn(t_in, self.linear1.weight, self.linear1.bias, self.linear2.weight, self.linear2.bias)
        if last or query_pos is None:
            return self.norm(t_res, y)  # :178-182
        return self.norm(t_res, y, fanout=2, pos=query_pos)


class _Scramble(torch.autograd.Function):
    """audio [BT, C] -> audio[idx] [BT, Q, C]; inv [BT, Q]: the flat (frame, query) positions that read frame f's token."""

    @staticmethod
    def forward(ctx, audio, idx, inv):
        ctx.save_for_backward(inv)
        return audio[idx]

    @staticmethod
    def backward(ctx, dy):
        (inv,) = ctx.saved_tensors
        return dy.reshape(-1, dy.shape[-1])[inv].sum(1), None, None


BATCH_CLASS_HEADS = True  # (tools/ab_const.py flips it for the A/B)


@TRANSFORMER_DECODER_REGISTRY.register()
class MultiScaleMaskedTransformerDecoder(nn.Module):
    _version = 2

    def _load_from_state_dict(self, state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys, error_msgs):
        version = local_metadata.get("version", None)
        if version is None or version < 2:  # transformer_decoder.py:226-245
            for k in list(state_dict.keys()):
                if "static_query" in k:
                    state_dict[k.replace("static_query", "query_feat")] = state_dict.pop(k)
                    logging.getLogger(__name__).warning("Weight format of %s have changed! Applying automatic conversion now ...",
                                                        self.__class__.__name__)
        super()._load_from_state_dict(state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys, error_msgs)

    def __init__(self, in_channels, mask_classification=True, *, num_classes: int, hidden_dim: int, num_queries: int,
                 num_frames: int, queries_fuse_type: str, audio_out_dim: int, nheads: int, dim_feedforward: int,
                 dec_layers: int, pre_norm: bool, mask_dim: int, enforce_input_project: bool, dataset_name: str,
                 use_cosine_loss: bool):
        super().__init__()
        assert mask_classification, "Only support mask classification model"
        self.mask_classification = mask_classification
        self.hidden_dim = hidden_dim
        self.num_heads, self.num_layers = nheads, dec_layers
        self.queries_fuse_type, self.audio_out_dim = queries_fuse_type, audio_out_dim
        self.transformer_self_attention_layers = nn.ModuleList()
        self.transformer_cross_attention_layers = nn.ModuleList()
        self.transformer_ffn_layers = nn.ModuleList()
        for _ in range(self.num_layers):
            self.transformer_self_attention_layers.append(SelfAttentionLayer(hidden_dim, nheads, 0.0, normalize_before=pre_norm))
            self.transformer_cross_attention_layers.append(CrossAttentionLayer(hidden_dim, nheads, 0.0, normalize_before=pre_norm))
            self.transformer_ffn_layers.append(FFNLayer(hidden_dim, dim_feedforward, 0.0, normalize_before=pre_norm))
        self.decoder_norm = _deferred_layer_norm(hidden_dim)  # applied 10x per forward: the deferred queue sums the uses
        self.num_queries = num_queries
        query_feat_dim = hidden_dim - audio_out_dim if queries_fuse_type == "dim" else hidden_dim
        self.query_feat = nn.Embedding(num_queries, query_feat_dim)
        self.query_embed = nn.Embedding(num_queries, hidden_dim)
        self.num_feature_levels = 3
        self.level_embed = nn.Embedding(self.num_feature_levels, hidden_dim)
        self.input_proj = nn.ModuleList()
        for _ in range(self.num_feature_levels):
            if in_channels != hidden_dim or enforce_input_project:
                conv = nn.Conv2d(in_channels, hidden_dim, kernel_size=1)
                nn.init.kaiming_uniform_(conv.weight, a=1)
                nn.init.constant_(conv.bias, 0)
                self.input_proj.append(conv)
            else:
                self.input_proj.append(nn.Sequential())
        self.class_embed = Linear(hidden_dim, num_classes + 1)  # own kernels incl. the 3-wide gradients (csrc/gemm_smallm.hip)
        self.mask_embed = MLP(hidden_dim, hidden_dim, mask_dim, 3)
        self.dataset_name = dataset_name
        self.use_cosine_loss = use_cosine_loss
        # test hook: list of ops.masklogit.PackedMask, one per prediction head #0.., used INSTEAD of the masks computed from the
        # logits (tests inject the reference's own masks so that a gradient comparison does not hinge on a near-zero cell)
        self.attn_mask_override = None
        # test hook: a list that receives the PackedMask of every prediction head of the next forward (what attn_mask_override
        # accepts: a second run with the first run's masks injected has no discrete choice left in the decoder)
        self.record_attn_masks = None
        # the class logits are not needed inside the layer loop (only the mask embedding is, for the next layer's attention
        # mask): with this switch the 10 heads' `class_embed` calls become ONE call on the stacked decoder outputs after the
        # loop - one forward, one dX, one dW launch instead of ten each, and no accumulation of ten weight / bias gradients
        self.batch_class_heads = BATCH_CLASS_HEADS
        self._dec_list = None

    @classmethod
    def from_config(cls, cfg, in_channels, mask_classification):
        assert cfg.MODEL.MASK_FORMER.DEC_LAYERS >= 1
        return dict(
            in_channels=in_channels, mask_classification=mask_classification,
            num_classes=cfg.MODEL.SEM_SEG_HEAD.NUM_CLASSES, hidden_dim=cfg.MODEL.MASK_FORMER.HIDDEN_DIM,
            num_queries=cfg.MODEL.MASK_FORMER.NUM_OBJECT_QUERIES, num_frames=cfg.MODEL.FUSE_CONFIG.NUM_FRAMES,
            queries_fuse_type=cfg.MODEL.FUSE_CONFIG.QUERIES_FUSE_TYPE, audio_out_dim=cfg.MODEL.FUSE_CONFIG.AUDIO_OUT_DIM,
            nheads=cfg.MODEL.MASK_FORMER.NHEADS, dim_feedforward=cfg.MODEL.MASK_FORMER.DIM_FEEDFORWARD,
            dec_layers=cfg.MODEL.MASK_FORMER.DEC_LAYERS - 1, pre_norm=cfg.MODEL.MASK_FORMER.PRE_NORM,
            enforce_input_project=cfg.MODEL.MASK_FORMER.ENFORCE_INPUT_PROJ, mask_dim=cfg.MODEL.SEM_SEG_HEAD.MASK_DIM,
            dataset_name=cfg.DATASETS.TRAIN[0][:5], use_cosine_loss=cfg.MODEL.MASK_FORMER.COSINE_WEIGHT > 0)

    def scramble_audio(self, audio_features, bt):
        """transformer_decoder.py:437 `audio.repeat(1,Q,1).reshape(Q,-1,C)` is NOT a transpose: query q of frame b
        gets the audio token of frame (q*BT+b)//Q.  Reproduced as an index gather, batch-first [BT,Q,C]."""
        Q = self.num_queries
        key = (bt, str(audio_features.device))
        idx = getattr(self, "_scramble_idx", {}).get(key)
        if idx is None:
            q = torch.arange(Q, device=audio_features.device)[None, :]
            b = torch.arange(bt, device=audio_features.device)[:, None]
            idx = torch.div(q * bt + b, Q, rounding_mode="floor")  # [BT,Q]
            # every frame's token is used by exactly Q (frame, query) pairs: their flat positions, frame by frame - the gather's
            # backward is then itself a gather + a sum over Q (deterministic; ATen's index backward sorts the 4 000 indices and
            # walks the segments: 72 us + the sort)
            inv = torch.argsort(idx.flatten(), stable=True).view(bt, Q)
            assert bool((idx.flatten()[inv] == torch.arange(bt, device=idx.device)[:, None]).all())
            idx = (idx, inv)
            if not hasattr(self, "_scramble_idx"):
                self._scramble_idx = {}
            self._scramble_idx[key] = idx
        return _Scramble.apply(audio_features[:, 0], idx[0], idx[1])  # [BT,Q,C]

    def forward(self, x, audio_features, mask_features, mask=None):
        bt, c_m, h_m, w_m = mask_features.shape
        assert len(x) == self.num_feature_levels
        del mask
        mf_tok = mask_features.permute(0, 2, 3, 1).reshape(bt, h_m * w_m, c_m)  # free for channels_last input
        src, pos, size_list = [], [], []
        for i in range(self.num_feature_levels):
            size_list.append(tuple(x[i].shape[-2:]))
            p = position_embedding_sine(1, x[i].shape[2], x[i].shape[3], x[i].device, self.hidden_dim // 2)
            pos.append(p.flatten(2).transpose(1, 2))  # [1,hw,C]
            s = add_channel_vector(self.input_proj[i](x[i]).flatten(2), self.level_embed.weight[i], 1)  # (ops/colsum.py)
            src.append(s.transpose(1, 2))  # [BT,hw,C]
        src_k = [s_ + p_ for s_, p_ in zip(src, pos)]  # key input of the cross-attention layers of a level: once, not per layer
        # ... and so are the k / v projections: layers l, l + 3, l + 6 read level l - one pair of GEMMs per level computes them
        # for all three (ops/linear.py memory_kv; their gradients come back through one [tokens, 3 x 256] buffer per level)
        nl = self.num_feature_levels
        attn = [layer.multihead_attn for layer in self.transformer_cross_attention_layers]
        kv = memory_kv(src_k, src, [[(a.in_proj_weight, a.in_proj_bias) for a in attn[l::nl]] for l in range(nl)], defer=True,
                       key_from_value=not any(p_.requires_grad for p_ in pos))  # src_k = src + (sine embedding): see memory_kv
        query_embed = self.query_embed.weight.unsqueeze(0)  # [1,Q,C]
        output = self.query_feat.weight.unsqueeze(0).expand(bt, -1, -1)
        if self.queries_fuse_type == "add":
            output = output + self.scramble_audio(audio_features, bt)
        elif self.queries_fuse_type == "dim":
            output = torch.cat([output, self.scramble_audio(audio_features, bt)], dim=-1)
        elif self.queries_fuse_type == "all":
            output = self.scramble_audio(audio_features, bt)
        from ..ops import masklogit
        nheads_pred = self.num_layers + 1
        # one buffer for the mask logits of all prediction heads ...
        logit_buf = torch.empty(nheads_pred, bt, self.num_queries, h_m * w_m, device=mf_tok.device, dtype=torch.float32)
        # Fused mask path (csrc/maskbits.hip): a layer's attention mask comes from the mask embedding and the pixel embedding
        # DOWNSAMPLED to the layer's memory size (interpolation and contraction commute), in one kernel whose MFMA result tile
        # is balloted into the bit-packed mask - the full-resolution logits are not needed inside the layer loop and are
        # computed for all heads by ONE launch after it.
        if not (mf_tok.is_cuda and mf_tok.dtype == torch.float32 and c_m == 256 and mf_tok.is_contiguous()
                and max(h * w for h, w in size_list) <= 4096):
            raise RuntimeError("MultiScaleMaskedTransformerDecoder: contiguous fp32 CUDA mask features with 256 channels and memory "
                               "levels of <= 4096 tokens expected (csrc/maskbits.hip; there is no CPU / generic fallback)")
        self._mfd = {sz: masklogit.downsample_tokens(mf_tok, (h_m, w_m), sz) for sz in set(size_list)}
        predictions_class, mask_embeds = [], []
        self._head_no = 0
        self._dec_list = [] if self.batch_class_heads else None
        outputs_class, mask_embed, blocked = self.forward_prediction_heads(output, mf_tok, (h_m, w_m), size_list[0], logit_buf[0])
        predictions_class.append(outputs_class)
        mask_embeds.append(mask_embed)
        fan = None  # handles on the previous layer's output (aliases: gradients are summed inside the LayerNorm backward kernel)
        for i in range(self.num_layers):
            lvl = i % self.num_feature_levels
            last = i == self.num_layers - 1
            # `blocked` already has the fully-blocked-row reset of :458 applied (fused in the mask kernel)
            fan = self.transformer_cross_attention_layers[i](output, src[lvl], blocked, pos[lvl], query_embed, src_k[lvl], fan=fan,
                                                             kv=kv[lvl][i // nl])
            fan = self.transformer_self_attention_layers[i](None, query_embed, fan=fan)
            out = self.transformer_ffn_layers[i](None, fan=fan, query_pos=query_embed, last=last)
            if last:
                output, fan = out, None
            else:
                output, fan = out[0], (out[1], out[2])
            outputs_class, mask_embed, blocked = self.forward_prediction_heads(
                output, mf_tok, (h_m, w_m), size_list[(i + 1) % self.num_feature_levels], logit_buf[i + 1])
            predictions_class.append(outputs_class)
            mask_embeds.append(mask_embed)
        assert len(predictions_class) == self.num_layers + 1
        if self._dec_list is not None:  # (:495 for all heads at once; rows are independent: the same values per head)
            predictions_class = list(self.class_embed(torch.stack(self._dec_list)).unbind(0))
            self._dec_list = None
        masklogit.mask_logits_all_into(mask_embeds, mf_tok, logit_buf)  # all heads' full-resolution logits: one launch
        self._mfd = {}
        # ... and ONE autograd node carries the gradient of all heads back to mask_features / the mask embeddings
        logits_all = masklogit.attach_mask_logit_grads(mf_tok, logit_buf, mask_embeds)
        parts = logits_all.unbind(0)  # ONE backward node (a stack of the 10 head gradients), not 19 zero-padded slices
        predictions_mask = [p.view(bt, self.num_queries, h_m, w_m) for p in parts]
        middles = list(parts[:-1]) if self.use_cosine_loss else []
        return {
            "pred_logits": predictions_class[-1], "pred_masks": predictions_mask[-1],
            "aux_outputs": [{"pred_logits": a, "pred_masks": b} for a, b in zip(predictions_class[:-1], predictions_mask[:-1])],
            "middles_attn_mask": middles,
            # the same logits as ONE tensor [heads, BT, Q, h, w] (head i = prediction after layer i-1, last = final): the
            # criterion's fused path reads and differentiates this buffer directly (ops/maskloss.py::mask_and_cosine)
            "_logits_all": logits_all.view(nheads_pred, bt, self.num_queries, h_m, w_m),
        }

    def forward_prediction_heads(self, output, mf_tok, hw, attn_mask_target_size, logits_out):
        """:493-509 -> (class logits [BT,Q,K+1], mask embedding [BT,Q,C], blocked bool [BT,Q,h*w]); the mask logits
        [BT,Q,H*W] are written into `logits_out` (their gradient is attached later by one node for all heads)."""
        from ..ops import masklogit
        if self._dec_list is not None and hasattr(self.decoder_norm, "defer_dw"):
            # two handles on the normalised output (class head, mask-embedding MLP): their gradients are summed inside the
            # LayerNorm backward kernel, not by autograd's accumulation (a clone + an add per head)
            dec_c, dec = self.decoder_norm(output, fanout=2)
        else:
            dec_c = dec = self.decoder_norm(output)
        if self._dec_list is not None:
            self._dec_list.append(dec_c)
            outputs_class = None  # filled in by forward() after the loop
        else:
            outputs_class = self.class_embed(dec_c)
        mask_embed = self.mask_embed(dec)
        # (the mask of the LAST head is never used - the reference computes it anyway, transformer_decoder.py:474-478)
        blocked = masklogit.mask_bits(mask_embed, self._mfd[tuple(attn_mask_target_size)]) if self._head_no < self.num_layers else None
        if self.record_attn_masks is not None and blocked is not None:
            self.record_attn_masks.append(blocked)
        if self.attn_mask_override is not None and self._head_no < len(self.attn_mask_override):
            blocked = self.attn_mask_override[self._head_no]
        self._head_no += 1
        return outputs_class, mask_embed, blocked
