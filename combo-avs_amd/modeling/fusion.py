"""Bilateral audio-visual fusion, mirroring fusion_module/AVFuse.py and
fusion_module/utils/fuse_helper.py of the reference: same class / parameter names and the same
`AVFuse(visual: dict, audio) -> {"visual": dict, "audio": tensor}` contract (MHA-B, late fusion).

MI355X-first restatement.  There is exactly ONE audio token per frame (maskformer_model.py:327-329), so
the score tensor is [BT*heads, HW, 1] and both softmaxes run over HW (fuse_helper.py:202-203).  With a single
key/value token the three [BT*HW,256]x[256,256] projections of the reference collapse algebraically:

    s[h,i]      = scale * (v_proj(xn_i + pos_i))_h . k_h  = (xn_i + pos_i) . u_h + c_h,
                  u_h = scale * Wq_h^T k_h  (256-d),  c_h = scale * bq_h . k_h
    out_v[i,:]  = W_ov (p[:,i] (x) va) + b_ov             = sum_h p[h,i] z_h + b_ov,   z_h = W_ov[:, h-block] va_h
    attn_a[h,:] = sum_i p[h,i] values_v_proj(xn_i)_h      = W_vv[h-block] (sum_i p[h,i] xn_i) + (sum_i p[h,i]) b_vv[h-block]

i.e. per frame one [HW,256]x[256,8] product, a softmax over HW per head, one [8,HW]x[HW,256] pooling and a
rank-8 update: ~13 MFLOP/frame instead of 1 238 MFLOP, purely HBM-bound.  Same arithmetic as the reference up
to fp32 re-association (parity test: tests/test_head_gpu.py).
"""
import torch
import torch.nn.functional as F
from torch import nn

from .layers import position_embedding_sine
from ..ops.colsum import add_channel_vector


class BiMultiHeadAttention(nn.Module):
    """fuse_helper.py:108-237 (parameters only; the math lives in BiAttentionBlock.fused_call)"""

    def __init__(self, v_dim, a_dim, embed_dim, num_heads, dropout=0.1):
        super().__init__()
        self.embed_dim, self.num_heads = embed_dim, num_heads
        self.head_dim = embed_dim // num_heads
        self.v_dim, self.a_dim = v_dim, a_dim
        assert self.head_dim * self.num_heads == self.embed_dim, \
            f"embed_dim must be divisible by num_heads (got `embed_dim`: {self.embed_dim} and `num_heads`: {self.num_heads})."
        self.scale = self.head_dim ** (-0.5)
        self.dropout = dropout
        self.v_proj = nn.Linear(self.v_dim, self.embed_dim)
        self.a_proj = nn.Linear(self.a_dim, self.embed_dim)
        self.values_v_proj = nn.Linear(self.v_dim, self.embed_dim)
        self.values_a_proj = nn.Linear(self.a_dim, self.embed_dim)
        self.out_v_proj = nn.Linear(self.embed_dim, self.v_dim)
        self.out_a_proj = nn.Linear(self.embed_dim, self.a_dim)
        self._reset_parameters()

    def _reset_parameters(self):
        for lin in (self.v_proj, self.a_proj, self.values_v_proj, self.values_a_proj, self.out_v_proj, self.out_a_proj):
            nn.init.xavier_uniform_(lin.weight)
            lin.bias.data.fill_(0)


class BiAttentionBlock(nn.Module):
    """fuse_helper.py:240-332"""

    def __init__(self, visual_features_names, vision_dim_list, audio_dim, embed_dim, num_heads, hidden_dim=None,
                 dropout=0.1, drop_path=0.0, init_values=1e-4):
        super().__init__()
        assert drop_path == 0.0
        self.visual_features_names = visual_features_names
        self.layer_norm_v_list = nn.ModuleList()
        self.layer_norm_a_list = nn.ModuleList()
        self.attn_list = nn.ModuleList()
        self.gamma_v_list = nn.ParameterList()
        for vision_dim in vision_dim_list:
            self.layer_norm_v_list.append(nn.LayerNorm(vision_dim))
            self.layer_norm_a_list.append(nn.LayerNorm(audio_dim))
            self.attn_list.append(BiMultiHeadAttention(v_dim=vision_dim, a_dim=audio_dim, embed_dim=embed_dim,
                                                       num_heads=num_heads, dropout=dropout))
            self.gamma_v_list.append(nn.Parameter(init_values * torch.ones((vision_dim)), requires_grad=True))
        self.gamma_a = nn.Parameter(init_values * torch.ones((audio_dim)), requires_grad=True)
        self.token_op = None  # set by AVFuse: the fused HIP implementation of the token stage

    def single_attention_call(self, v_tok, a, level, pos_v, pos_a):
        """v_tok [B,N,C] token-major (NOT yet LayerNormed, level_embed already added), a [B,1,Ca].
        Returns (new_v [B,N,C], new_a [B,1,Ca])   (fuse_helper.py:320-332, 155-237)."""
        attn = self.attn_list[level]
        H, hd = attn.num_heads, attn.head_dim
        ln_v, ln_a = self.layer_norm_v_list[level], self.layer_norm_a_list[level]
        B = v_tok.shape[0]
        a_ln = ln_a(a)  # [B,1,Ca]
        k = attn.a_proj(a_ln + pos_a).view(B, H, hd)
        va = attn.values_a_proj(a_ln).view(B, H, hd)
        Wq = attn.v_proj.weight.view(H, hd, -1)
        u = torch.einsum("bhd,hdc->bhc", k, Wq) * attn.scale
        c = torch.einsum("bhd,hd->bh", k, attn.v_proj.bias.view(H, hd)) * attn.scale
        z = torch.einsum("chd,bhd->bhc", attn.out_v_proj.weight.view(-1, H, hd), va)
        p_drop = attn.dropout if self.training else 0.0
        y, pooled, spa = self.token_op(v_tok, ln_v.weight, ln_v.bias, ln_v.eps, pos_v, u, c, z, attn.out_v_proj.bias,
                                       self.gamma_v_list[level], p_drop)
        attn_a = torch.einsum("bhc,hdc->bhd", pooled, attn.values_v_proj.weight.view(H, hd, -1)) \
            + spa[..., None] * attn.values_v_proj.bias.view(H, hd)
        out_a = attn.out_a_proj(attn_a.reshape(B, 1, H * hd))
        return y, a_ln + self.gamma_a * out_a

    def forward(self, visual_features, audio_feature, pos_a=None, pos_v=None):
        new_a_list = []
        for level, name in enumerate(self.visual_features_names):
            feat = visual_features[name]
            bs, c, h, w = feat.shape
            v_tok = feat.permute(0, 2, 3, 1).reshape(bs, h * w, c)  # free if feat is channels_last
            new_v, new_a = self.single_attention_call(v_tok, audio_feature, level, pos_v, pos_a)
            visual_features[name] = new_v.view(bs, h, w, c).permute(0, 3, 1, 2)  # NCHW view of token-major memory
            new_a_list.append(new_a)
        audio_feature = torch.mean(torch.stack(new_a_list, dim=1), dim=1)
        return visual_features, audio_feature


class AVFuse(nn.Module):
    """AVFuse.py:10-125 (fused_type 'MHA-B' - the only type the shipped configs use - and 'MHA-None')."""

    def __init__(self, fused_type, audio_dim, fused_backbone, fused_backbone_dim):
        super().__init__()
        self.fused_type, self.audio_dim = fused_type, audio_dim
        self.fused_backbone, self.fused_backbone_dim = list(fused_backbone), list(fused_backbone_dim)
        self.n_head = 8
        self.embed_dim = max(self.fused_backbone_dim)
        self.hidden_dim = self.embed_dim * 4
        self.audio_pos = nn.Embedding(1, self.audio_dim)
        self.level_embed = nn.Embedding(1, self.fused_backbone_dim[0])
        if self.fused_type == "MHA-B":
            self.b_attn = BiAttentionBlock(visual_features_names=self.fused_backbone,
                                           vision_dim_list=self.fused_backbone_dim, audio_dim=self.audio_dim,
                                           embed_dim=self.embed_dim, num_heads=self.n_head,
                                           hidden_dim=self.hidden_dim, dropout=0.1, drop_path=0.0)
            from ..ops import bifuse
            self.b_attn.token_op = bifuse.token_op
        elif self.fused_type != "MHA-None":
            raise NotImplementedError(f"fusion type {self.fused_type!r}: only MHA-B / MHA-None are shipped by the reference configs")

    @classmethod
    def from_config(cls, cfg):
        return dict(fused_type=cfg.MODEL.FUSE_CONFIG.TYPE, audio_dim=cfg.MODEL.FUSE_CONFIG.AUDIO_DIM,
                    fused_backbone=cfg.MODEL.FUSE_CONFIG.FUSED_BACKBONE,
                    fused_backbone_dim=cfg.MODEL.FUSE_CONFIG.FUSED_BACKBONE_DIM)

    def forward(self, visual_features, audio_features):
        if self.fused_type == "MHA-None":
            return {"visual": visual_features, "audio": audio_features}
        audio_pos = self.audio_pos.weight.unsqueeze(0)  # [1,1,Ca], broadcast over frames (AVFuse.py:97-98)
        image_pos = None
        for i, name in enumerate(self.fused_backbone):
            f = visual_features[name]
            # the LAST level's PE is what reaches b_attn (AVFuse.py:103,109); one level in every shipped config
            image_pos = position_embedding_sine(1, f.shape[2], f.shape[3], f.device, self.fused_backbone_dim[0] // 2)
            image_pos = image_pos.flatten(2).permute(0, 2, 1)  # [1,HW,C]
            # AVFuse.py:104-106; the embedding's gradient from csrc/colsum.hip (ops/colsum.py: no ATen multi-workgroup reduction)
            visual_features[name] = add_channel_vector(f, self.level_embed.weight[i], 1)
        v, a = self.b_attn(visual_features, audio_features, pos_v=image_pos, pos_a=audio_pos)
        return {"visual": v, "audio": a}
