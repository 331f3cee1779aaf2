"""CPU ORACLE for the COMBO-AVS fusion + mask-decoding hot path.  TEST INFRASTRUCTURE ONLY.

This is a from-scratch, functional, plain-PyTorch (CPU, fp32/fp64) *restatement* of the reference's
algorithm for the path SURVEY.md §8(a) lists (rows a1-a17).  It is not the product: only `tests/`,
`__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import it; the shipped package
(`combo-avs_amd/`) never does and fails loudly when its HIP library is missing.

Parity pin: every function below is checked against golden vectors produced by *running the
reference's own Python* in the build container (`tests/golden/gen_golden.py` -> `tests/golden/*.npz`,
checked by `tests/test_oracle_golden.py`).  The reference's only known-answer test for this path is
`models/modeling/pixel_decoder/ops/test.py` (a relation test, seed 3); its three cases are among the
fixtures.  Third-party arithmetic whose source is not in /root/reference (detectron2 0.6 `Conv2d`,
`point_sample`, `get_uncertain_point_coords_with_randomness`, `ImageList`; torch's
`nn.MultiheadAttention`/`grid_sample`) is restated from its published behaviour; the backbones
(d2 ResNet-50, out of the hot path) are "parity unpinned" because detectron2 is not available here.

Parameters are passed as a flat dict `P` keyed by the reference's state-dict names (SURVEY.md §8(b)
"checkpoint surface"), `pre` is the key prefix of the sub-module.
All citations are `path:line` under /root/reference.
"""
import math

import torch
import torch.nn.functional as F

# =====================================================================================================
# a3  PositionEmbeddingSine                         models/modeling/transformer_decoder/position_encoding.py:29-48
# =====================================================================================================

def position_embedding_sine(b, h, w, num_pos_feats=128, temperature=10000.0, scale=2 * math.pi, dtype=torch.float32):
    """normalize=True variant (the only one used: msdeformattn.py:240, AVFuse.py:35, transformer_decoder.py:306)."""
    eps = 1e-6
    y = torch.arange(1, h + 1, dtype=torch.float32).view(h, 1).expand(h, w)  # cumsum of ones, :35
    x = torch.arange(1, w + 1, dtype=torch.float32).view(1, w).expand(h, w)  # :36
    y = y / (float(h) + eps) * scale  # :39
    x = x / (float(w) + eps) * scale  # :40
    i = torch.arange(num_pos_feats, dtype=torch.float32)
    dim_t = temperature ** (2 * torch.div(i, 2, rounding_mode="floor") / num_pos_feats)  # :42-43
    px = x[:, :, None] / dim_t
    py = y[:, :, None] / dim_t
    px = torch.stack((px[:, :, 0::2].sin(), px[:, :, 1::2].cos()), dim=3).flatten(2)  # :47
    py = torch.stack((py[:, :, 0::2].sin(), py[:, :, 1::2].cos()), dim=3).flatten(2)  # :48
    pos = torch.cat((py, px), dim=2).permute(2, 0, 1)  # [2*npf, h, w]
    return pos.unsqueeze(0).expand(b, -1, -1, -1).to(dtype)


# =====================================================================================================
# a1  Siam-Encoder-Module mix          models/utils/misc.py:112-131 + models/maskformer_model.py:345-352
# =====================================================================================================

def channel_weighted_gate(P, pre, p):
    """SE-style gate s = sigmoid(W2 relu(W1 GAP(p)))  -> [B,C,1,1]   (misc.py:123-131)"""
    y = p.mean(dim=(2, 3))
    y = F.relu(F.linear(y, P[pre + "fc1.weight"], P[pre + "fc1.bias"]))
    y = torch.sigmoid(F.linear(y, P[pre + "fc2.weight"], P[pre + "fc2.bias"]))
    return y[:, :, None, None]


def sem_mix(P, pre, features, pre_sam_features):
    """f_l <- f_l + gate_l(p_l) * p_l for the 4 levels (maskformer_model.py:345-352);
    `pre` is the prefix of `scale_factor_module.` (ModuleList index = level)."""
    out = {}
    for i, k in enumerate(features.keys()):
        s = channel_weighted_gate(P, f"{pre}{i}.", pre_sam_features[k])
        out[k] = features[k] + s * pre_sam_features[k]
    return out


# =====================================================================================================
# a6  MSDeformAttn core op
#     oracle of record in the reference: ops/functions/ms_deform_attn_func.py:53-72 (grid_sample)
#     native semantics restated here:    ops/src/cuda/ms_deform_im2col_cuda.cuh:38-89, 242-304
# =====================================================================================================

def ms_deform_attn_core(value, spatial_shapes, sampling_locations, attention_weights):
    """out[b,q,m,:] = sum_{l,p} w[b,q,m,l,p] * bilinear(value_l[b,:,m,:], loc*(W_l,H_l) - 0.5), zeros outside.

    Written as explicit 4-tap gathers (the .cuh formulation, `h_im = loc_h*H - 0.5`, taps with
    index outside [0,H-1]x[0,W-1] contribute 0) rather than through grid_sample, so that it is an
    independent statement of the op; differentiable w.r.t. value / locations / weights via autograd.
    value [B,S,M,D]; spatial_shapes [[H,W]]*L (ints); loc [B,Lq,M,L,P,2] (x,y in [0,1]); w [B,Lq,M,L,P].
    """
    B, S, M, D = value.shape
    _, Lq, _, L, Pn, _ = sampling_locations.shape
    shapes = [(int(h), int(w)) for h, w in (spatial_shapes.tolist() if torch.is_tensor(spatial_shapes) else spatial_shapes)]
    out = value.new_zeros(B, Lq, M, D)
    start = 0
    for lid, (H, W) in enumerate(shapes):
        v = value[:, start:start + H * W]  # [B,HW,M,D]
        start += H * W
        v = v.permute(0, 2, 1, 3)  # [B,M,HW,D]
        loc = sampling_locations[:, :, :, lid]  # [B,Lq,M,P,2]
        x = loc[..., 0] * W - 0.5  # w_im  (.cuh:291)
        y = loc[..., 1] * H - 0.5  # h_im  (.cuh:290)
        x0 = torch.floor(x)
        y0 = torch.floor(y)
        lw = x - x0
        lh = y - y0
        acc = 0
        for dy, dx, wt in ((0, 0, (1 - lh) * (1 - lw)), (0, 1, (1 - lh) * lw), (1, 0, lh * (1 - lw)), (1, 1, lh * lw)):
            yi = y0 + dy
            xi = x0 + dx
            ok = (yi >= 0) & (yi <= H - 1) & (xi >= 0) & (xi <= W - 1)
            idx = (yi.clamp(0, H - 1) * W + xi.clamp(0, W - 1)).long()  # [B,Lq,M,P]
            idx = idx.permute(0, 2, 1, 3).reshape(B, M, Lq * Pn)  # [B,M,Lq*P]
            g = torch.gather(v, 2, idx[..., None].expand(-1, -1, -1, D)).view(B, M, Lq, Pn, D)
            g = g.permute(0, 2, 1, 3, 4)  # [B,Lq,M,P,D]
            acc = acc + g * (wt * ok.to(wt.dtype))[..., None]
        out = out + (acc * attention_weights[:, :, :, lid][..., None]).sum(3)
    return out.reshape(B, Lq, M * D)


# =====================================================================================================
# a5  MSDeformAttn module                          ops/modules/ms_deform_attn.py:86-129
# =====================================================================================================

def ms_deform_attn(P, pre, query, reference_points, input_flatten, spatial_shapes, n_heads=8, n_points=4):
    N, Lq, C = query.shape
    _, Lin, _ = input_flatten.shape
    L = len(spatial_shapes)
    value = F.linear(input_flatten, P[pre + "value_proj.weight"], P[pre + "value_proj.bias"])  # :102
    value = value.view(N, Lin, n_heads, C // n_heads)  # :105
    off = F.linear(query, P[pre + "sampling_offsets.weight"], P[pre + "sampling_offsets.bias"])
    off = off.view(N, Lq, n_heads, L, n_points, 2)  # :106
    aw = F.linear(query, P[pre + "attention_weights.weight"], P[pre + "attention_weights.bias"])
    aw = F.softmax(aw.view(N, Lq, n_heads, L * n_points), -1).view(N, Lq, n_heads, L, n_points)  # :107-108
    normalizer = torch.tensor([[w, h] for h, w in spatial_shapes], dtype=query.dtype)  # (W,H) order, :111
    loc = reference_points[:, :, None, :, None, :] + off / normalizer[None, None, None, :, None, :]  # :112
    out = ms_deform_attn_core(value, spatial_shapes, loc, aw)  # :119-125
    return F.linear(out, P[pre + "output_proj.weight"], P[pre + "output_proj.bias"])  # :128


# =====================================================================================================
# a4  deformable encoder                           pixel_decoder/msdeformattn.py:67-95, 119-134, 144-165
# =====================================================================================================

def encoder_reference_points(spatial_shapes, batch, dtype=torch.float32):
    """msdeformattn.py:144-157 with valid_ratios == 1 (masks all False, :68)."""
    pts = []
    for (H, W) in spatial_shapes:
        ry = (torch.arange(H, dtype=dtype) + 0.5) / H  # linspace(0.5, H-0.5, H) / H
        rx = (torch.arange(W, dtype=dtype) + 0.5) / W
        gy, gx = torch.meshgrid(ry, rx, indexing="ij")
        pts.append(torch.stack((gx.reshape(-1), gy.reshape(-1)), -1))
    ref = torch.cat(pts, 0)  # [S,2] (x,y)
    L = len(spatial_shapes)
    return ref[None, :, None, :].expand(batch, -1, L, -1)


def layer_norm(P, pre, x, eps=1e-5):
    return F.layer_norm(x, (x.shape[-1],), P[pre + "weight"], P[pre + "bias"], eps)


def encoder_layer(P, pre, src, pos, ref, spatial_shapes):
    """post-norm layer, dropout p = MASK_FORMER.DROPOUT = 0 (msdeformattn.py:119-134)."""
    src2 = ms_deform_attn(P, pre + "self_attn.", src + pos, ref, src, spatial_shapes)
    src = layer_norm(P, pre + "norm1.", src + src2)
    ff = F.linear(F.relu(F.linear(src, P[pre + "linear1.weight"], P[pre + "linear1.bias"])),
                  P[pre + "linear2.weight"], P[pre + "linear2.bias"])
    return layer_norm(P, pre + "norm2.", src + ff)


def deformable_encoder(P, pre, srcs, pos_embeds, num_layers=6):
    """MSDeformAttnTransformerEncoderOnly.forward (msdeformattn.py:67-95); `pre` = '...pixel_decoder.transformer.'"""
    spatial_shapes = [(s.shape[2], s.shape[3]) for s in srcs]
    src = torch.cat([s.flatten(2).transpose(1, 2) for s in srcs], 1)
    pos = torch.cat([p.flatten(2).transpose(1, 2) + P[pre + "level_embed"][l].view(1, 1, -1)
                     for l, p in enumerate(pos_embeds)], 1)  # :80-82
    ref = encoder_reference_points(spatial_shapes, src.shape[0], src.dtype)
    for i in range(num_layers):
        src = encoder_layer(P, f"{pre}encoder.layers.{i}.", src, pos, ref, spatial_shapes)
    return src, spatial_shapes


# =====================================================================================================
# a2  MSDeformAttnPixelDecoder.forward_features    pixel_decoder/msdeformattn.py:315-359
# =====================================================================================================

def group_norm(P, pre, x, groups=32, eps=1e-5):
    return F.group_norm(x, groups, P[pre + "weight"], P[pre + "bias"], eps)


def pixel_decoder_forward_features(P, pre, features, num_layers=6):
    """-> mask_features [BT,256,H/4,W/4], out[0], multi_scale = [res5-, res4-, res3-resolution maps]"""
    srcs, pos = [], []
    for idx, f in enumerate(("res5", "res4", "res3")):  # :320  (input_proj index 0 <-> res5, :215-224)
        x = features[f].float()
        y = F.conv2d(x, P[f"{pre}input_proj.{idx}.0.weight"], P[f"{pre}input_proj.{idx}.0.bias"])
        srcs.append(group_norm(P, f"{pre}input_proj.{idx}.1.", y))
        pos.append(position_embedding_sine(x.shape[0], x.shape[2], x.shape[3]))
    y, spatial_shapes = deformable_encoder(P, pre + "transformer.", srcs, pos, num_layers)  # :325
    bs = y.shape[0]
    out, start = [], 0
    for (H, W) in spatial_shapes:  # :328-340
        out.append(y[:, start:start + H * W].transpose(1, 2).reshape(bs, -1, H, W))
        start += H * W
    # one FPN level (res2): lateral 1x1+GN (no bias), bilinear up, 3x3+GN+ReLU      :344-352
    x = features["res2"].float()
    cur = group_norm(P, pre + "adapter_1.norm.", F.conv2d(x, P[pre + "adapter_1.weight"]))
    y = cur + F.interpolate(out[-1], size=cur.shape[-2:], mode="bilinear", align_corners=False)
    y = F.relu(group_norm(P, pre + "layer_1.norm.", F.conv2d(y, P[pre + "layer_1.weight"], padding=1)))
    out.append(y)
    mask_features = F.conv2d(out[-1], P[pre + "mask_features.weight"], P[pre + "mask_features.bias"])  # :359
    return mask_features, out[0], out[:3]


# =====================================================================================================
# a7-a9  AVFuse / BiAttentionBlock / BiMultiHeadAttention
#        fusion_module/AVFuse.py:92-125, fusion_module/utils/fuse_helper.py:155-237, 290-332
# =====================================================================================================

def bi_multihead_attention(P, pre, v, a, pos_v, pos_a, num_heads=8, dropout_masks=None):
    """fuse_helper.py:155-237.  v [B,HW,Cv] (already LayerNormed), a [B,1,Ca].
    With ONE audio token the score tensor is [B*heads, HW, 1] and BOTH softmaxes run over HW
    (:202 softmax(dim=-1) of the transposed scores, :203 softmax(dim=1)).
    dropout_masks: optional (mask_v, mask_a) of shape [B*heads,HW,1]/[B*heads,1,HW], already scaled by
    1/(1-p), to emulate train-mode dropout (:204-205) deterministically."""
    B, HW, _ = v.shape
    E = P[pre + "v_proj.weight"].shape[0]
    hd = E // num_heads
    scale = hd ** (-0.5)
    q = F.linear(v + pos_v, P[pre + "v_proj.weight"], P[pre + "v_proj.bias"]) * scale  # :161
    k = F.linear(a + pos_a, P[pre + "a_proj.weight"], P[pre + "a_proj.bias"])  # :166
    vv = F.linear(v, P[pre + "values_v_proj.weight"], P[pre + "values_v_proj.bias"])  # :168
    va = F.linear(a, P[pre + "values_a_proj.weight"], P[pre + "values_a_proj.bias"])  # :169

    def heads(t):  # _shape + view(proj_shape)  (:146-147, :171-176)
        return t.view(B, -1, num_heads, hd).transpose(1, 2).reshape(B * num_heads, -1, hd)
    q, k, vv, va = heads(q), heads(k), heads(vv), heads(va)
    s = torch.bmm(q, k.transpose(1, 2))  # [B*h, HW, 1]  :180
    s = s.clamp(min=-50000, max=50000)  # :190-193
    sT = s.transpose(1, 2)
    sa = (sT - sT.max(dim=-1, keepdim=True)[0]).clamp(min=-50000, max=50000)  # :197-201
    p_a = sa.softmax(dim=-1)  # [B*h,1,HW]  :202
    p_v = F.softmax(s, dim=1)  # [B*h,HW,1]  :203
    if dropout_masks is not None:
        p_v = p_v * dropout_masks[0]
        p_a = p_a * dropout_masks[1]
    out_v = torch.bmm(p_v, va)  # rank-1: p[hw] * va   :207-209
    out_a = torch.bmm(p_a, vv)  # attention pooling     :210-212
    out_v = out_v.view(B, num_heads, HW, hd).transpose(1, 2).reshape(B, HW, E)  # :226-228
    out_a = out_a.view(B, num_heads, 1, hd).transpose(1, 2).reshape(B, 1, E)  # :230-232
    out_v = F.linear(out_v, P[pre + "out_v_proj.weight"], P[pre + "out_v_proj.bias"])  # :234
    out_a = F.linear(out_a, P[pre + "out_a_proj.weight"], P[pre + "out_a_proj.bias"])  # :235
    return out_v, out_a, p_a


def avfuse(P, pre, mask_features, audio, dropout_masks=None, return_probs=False):
    """AVFuse.__call__ for the only shipped configuration (late fusion, one level 'res2', MHA-B).
    `pre` = '...fusion_module.'.  Returns (fused visual [BT,C,H,W], fused audio [BT,1,128])."""
    BT, C, H, W = mask_features.shape
    audio_pos = P[pre + "audio_pos.weight"].unsqueeze(0).expand(BT, -1, -1)  # AVFuse.py:97-98
    image_pos = position_embedding_sine(BT, H, W, C // 2).flatten(2).permute(0, 2, 1)  # :103
    x = mask_features + P[pre + "level_embed.weight"][0][None, :, None, None]  # :104-106
    v = x.permute(0, 2, 3, 1).reshape(BT, H * W, C)  # permute_and_flatten, fuse_helper.py:10-14
    b = pre + "b_attn."
    v = layer_norm(P, b + "layer_norm_v_list.0.", v)  # fuse_helper.py:326
    a = layer_norm(P, b + "layer_norm_a_list.0.", audio)  # :327
    dv, da, p = bi_multihead_attention(P, b + "attn_list.0.", v, a, image_pos, audio_pos, 8, dropout_masks)
    v = v + P[b + "gamma_v_list.0"] * dv  # residual on the LayerNormed tensor, :330
    a = a + P[b + "gamma_a"] * da  # :331
    fused_v = v.transpose(1, 2).reshape(BT, C, H, W)  # :305, :314
    if return_probs:
        return fused_v, a, p
    return fused_v, a  # mean over a single level is the identity, :309-310


# a10  audio_mlp                                       modeling/misc/audio_transformation.py:5-14
def audio_mlp(P, pre, x):
    e = pre + "embeddings."
    x = F.relu(F.linear(x, P[e + "0.weight"], P[e + "0.bias"]))
    x = F.relu(F.linear(x, P[e + "2.weight"], P[e + "2.bias"]))
    return F.linear(x, P[e + "4.weight"], P[e + "4.bias"])


# =====================================================================================================
# a11-a13  MultiScaleMaskedTransformerDecoder       transformer_decoder/transformer_decoder.py:405-509
# =====================================================================================================

def multihead_attention(P, pre, query, key, value, attn_mask=None, num_heads=8):
    """torch.nn.MultiheadAttention restated (seq-first [L,B,E], packed in_proj [3E,E], bool mask True=blocked,
    [B*heads, Lq, Lk] with batch index b*heads+h); used at transformer_decoder.py:30,82."""
    Lq, B, E = query.shape
    Lk = key.shape[0]
    hd = E // num_heads
    Wi, bi = P[pre + "in_proj_weight"], P[pre + "in_proj_bias"]
    q = F.linear(query, Wi[:E], bi[:E])
    k = F.linear(key, Wi[E:2 * E], bi[E:2 * E])
    v = F.linear(value, Wi[2 * E:], bi[2 * E:])
    q = q.reshape(Lq, B * num_heads, hd).transpose(0, 1) * (hd ** -0.5)
    k = k.reshape(Lk, B * num_heads, hd).transpose(0, 1)
    v = v.reshape(Lk, B * num_heads, hd).transpose(0, 1)
    s = torch.bmm(q, k.transpose(1, 2))
    if attn_mask is not None:
        s = s.masked_fill(attn_mask, float("-inf"))
    p = F.softmax(s, dim=-1)
    o = torch.bmm(p, v).transpose(0, 1).reshape(Lq, B, E)
    return F.linear(o, P[pre + "out_proj.weight"], P[pre + "out_proj.bias"])


def mlp3(P, pre, x):
    """MLP(256,256,mask_dim,3)  transformer_decoder.py:207-219"""
    x = F.relu(F.linear(x, P[pre + "layers.0.weight"], P[pre + "layers.0.bias"]))
    x = F.relu(F.linear(x, P[pre + "layers.1.weight"], P[pre + "layers.1.bias"]))
    return F.linear(x, P[pre + "layers.2.weight"], P[pre + "layers.2.bias"])


def forward_prediction_heads(P, pre, output, mask_features, target_size, num_heads=8):
    """transformer_decoder.py:493-509 -> (class logits [BT,Q,K+1], mask logits [BT,Q,H,W], bool mask [BT*heads,Q,hw])"""
    dec = layer_norm(P, pre + "decoder_norm.", output).transpose(0, 1)  # :494-495
    cls = F.linear(dec, P[pre + "class_embed.weight"], P[pre + "class_embed.bias"])  # :496
    me = mlp3(P, pre + "mask_embed.", dec)  # :497
    masks = torch.einsum("bqc,bchw->bqhw", me, mask_features)  # :498
    am = F.interpolate(masks, size=target_size, mode="bilinear", align_corners=False)  # :502
    am = (am.sigmoid().flatten(2).unsqueeze(1).repeat(1, num_heads, 1, 1).flatten(0, 1) < 0.5).bool()  # :504
    return cls, masks, am.detach()


def scramble_audio(audio, num_queries):
    """transformer_decoder.py:437: `audio.repeat(1,Q,1).reshape(Q,-1,C)` - NOT a transpose.
    Query q of frame b receives the audio token of frame (q*BT+b)//Q."""
    return audio.repeat(1, num_queries, 1).reshape(num_queries, -1, audio.shape[-1])


def transformer_decoder(P, pre, x, audio, mask_features, num_layers=9, num_heads=8, attn_override=None):
    """MultiScaleMaskedTransformerDecoder.forward (transformer_decoder.py:405-491); `pre` = '...predictor.'.
    x: 3 maps (7^2,14^2,28^2 at 224 input), audio [BT,1,256], mask_features [BT,256,H,W].
    Returns dict(pred_logits, pred_masks, aux_outputs, middles_attn_mask) + 'attn_masks' (the 10 bool masks).
    attn_override: list of bool [BT,Q,hw] masks AS PRODUCED by heads #0.. (before the row reset of :458) that replace the
    computed ones - test hook that freezes the decoder's discrete choices (golden `dec/attn_bits*`)."""
    def frozen(k, am):
        if attn_override is None or k >= len(attn_override):
            return am
        o = attn_override[k].unsqueeze(1).repeat(1, num_heads, 1, 1).flatten(0, 1)
        assert o.shape == am.shape
        return o
    bt = mask_features.shape[0]
    Q = P[pre + "query_feat.weight"].shape[0]
    src, pos, sizes = [], [], []
    for i in range(3):  # :419-426 (input_proj is an empty Sequential because in_channels == hidden_dim, :353-357)
        sizes.append(tuple(x[i].shape[-2:]))
        pos.append(position_embedding_sine(bt, *sizes[-1]).flatten(2).permute(2, 0, 1))
        src.append((x[i].flatten(2) + P[pre + "level_embed.weight"][i][None, :, None]).permute(2, 0, 1))
    query_embed = P[pre + "query_embed.weight"].unsqueeze(1).repeat(1, bt, 1)  # :431-432
    output = P[pre + "query_feat.weight"].unsqueeze(1).repeat(1, bt, 1)  # :433-434
    output = output + scramble_audio(audio, Q)  # QUERIES_FUSE_TYPE == "add", :437-440
    classes, masks, attn_masks, attn_used, middles = [], [], [], [], []
    c, m, am = forward_prediction_heads(P, pre, output, mask_features, sizes[0], num_heads)  # :451
    am = frozen(0, am)
    classes.append(c); masks.append(m); attn_masks.append(am)
    middles.append(m.reshape(bt, Q, -1))  # :455  (raw mask LOGITS, despite the name)
    for i in range(num_layers):
        lvl = i % 3
        am = am.clone()
        am[torch.where(am.sum(-1) == am.shape[-1])] = False  # :458  fully-blocked rows are unblocked
        attn_used.append(am)
        ca = f"{pre}transformer_cross_attention_layers.{i}."
        t2 = multihead_attention(P, ca + "multihead_attn.", output + query_embed, src[lvl] + pos[lvl], src[lvl], am, num_heads)
        output = layer_norm(P, ca + "norm.", output + t2)  # :99-118
        sa = f"{pre}transformer_self_attention_layers.{i}."
        qk = output + query_embed
        t2 = multihead_attention(P, sa + "self_attn.", qk, qk, output, None, num_heads)
        output = layer_norm(P, sa + "norm.", output + t2)  # :50-58
        ff = f"{pre}transformer_ffn_layers.{i}."
        t2 = F.linear(F.relu(F.linear(output, P[ff + "linear1.weight"], P[ff + "linear1.bias"])),
                      P[ff + "linear2.weight"], P[ff + "linear2.bias"])
        output = layer_norm(P, ff + "norm.", output + t2)  # :178-182
        c, m, am = forward_prediction_heads(P, pre, output, mask_features, sizes[(i + 1) % 3], num_heads)  # :474
        am = frozen(i + 1, am)
        classes.append(c); masks.append(m); attn_masks.append(am)
        if i != num_layers - 1:  # :479-482
            middles.append(m.reshape(bt, Q, -1))
    return {
        "pred_logits": classes[-1], "pred_masks": masks[-1],
        "aux_outputs": [{"pred_logits": a, "pred_masks": b} for a, b in zip(classes[:-1], masks[:-1])],
        "middles_attn_mask": middles, "attn_masks": attn_masks, "attn_masks_used": attn_used,
    }


# MaskFormerHead.layers                               meta_arch/mask_former_head.py:141-159
def head_forward(P, pre, features, audio, enc_layers=6, dec_layers=9, return_intermediates=False, attn_override=None):
    mf, _, ms = pixel_decoder_forward_features(P, pre + "pixel_decoder.", features, enc_layers)
    fv, fa = avfuse(P, pre + "fusion_module.", mf, audio)
    a256 = audio_mlp(P, pre + "audio_transformation.", fa)
    out = transformer_decoder(P, pre + "predictor.", ms, a256, fv, dec_layers, attn_override=attn_override)
    if return_intermediates:
        out["_inter"] = {"mask_features": mf, "multi_scale": ms, "fused_visual": fv, "fused_audio": fa, "audio256": a256}
    return out


# =====================================================================================================
# a14  HungarianMatcher                                modeling/matcher.py:84-136
# =====================================================================================================

def point_sample(x, coords):
    """detectron2 point_sample == grid_sample(x, 2c-1, bilinear, zeros, align_corners=False), restated
    as explicit 4-tap gathers.  x [N,C,H,W], coords [N,P,2] (x,y) in [0,1] -> [N,C,P]."""
    N, C, H, W = x.shape
    px = coords[..., 0] * W - 0.5
    py = coords[..., 1] * H - 0.5
    x0, y0 = torch.floor(px), torch.floor(py)
    lw, lh = px - x0, py - y0
    flat = x.reshape(N, C, H * W)
    out = 0
    for dy, dx, wt in ((0, 0, (1 - lh) * (1 - lw)), (0, 1, (1 - lh) * lw), (1, 0, lh * (1 - lw)), (1, 1, lh * lw)):
        yi, xi = y0 + dy, x0 + dx
        ok = (yi >= 0) & (yi <= H - 1) & (xi >= 0) & (xi <= W - 1)
        idx = (yi.clamp(0, H - 1) * W + xi.clamp(0, W - 1)).long()
        g = torch.gather(flat, 2, idx[:, None, :].expand(-1, C, -1))
        out = out + g * (wt * ok.to(wt.dtype))[:, None, :]
    return out


def matcher_cost(pred_logits_b, pred_masks_b, labels, gt_masks, point_coords, w_class=2.0, w_mask=5.0, w_dice=5.0):
    """Cost matrix [Q,G] for one frame (matcher.py:93-131); point_coords [1,P,2] shared by all masks (:107)."""
    prob = pred_logits_b.softmax(-1)
    cost_class = -prob[:, labels]  # :97
    G = gt_masks.shape[0]
    t = point_sample(gt_masks[:, None].to(pred_masks_b.dtype), point_coords.repeat(G, 1, 1)).squeeze(1)  # :109-113
    o = point_sample(pred_masks_b[:, None], point_coords.repeat(pred_masks_b.shape[0], 1, 1)).squeeze(1)  # :115-119
    o, t = o.float(), t.float()
    hw = o.shape[1]
    pos = F.softplus(-o)  # BCE(x, 1)   matcher.py:47
    neg = F.softplus(o)  # BCE(x, 0)   :48
    cost_mask = (pos @ t.T + neg @ (1 - t).T) / hw  # :50-52
    s = o.sigmoid()
    cost_dice = 1 - (2 * (s @ t.T) + 1) / (s.sum(-1)[:, None] + t.sum(-1)[None, :] + 1)  # :23-27
    return w_mask * cost_mask + w_class * cost_class + w_dice * cost_dice  # :131


def hungarian_matcher(pred_logits, pred_masks, targets, num_points=12544, rand=torch.rand):
    """-> list of (src_idx, tgt_idx) int64 tensors.  LSAP by scipy.optimize.linear_sum_assignment
    (the reference's own dependency, matcher.py:7,134)."""
    from scipy.optimize import linear_sum_assignment
    idx = []
    with torch.no_grad():
        for b in range(pred_logits.shape[0]):
            pc = rand(1, num_points, 2)
            C = matcher_cost(pred_logits[b], pred_masks[b], targets[b]["labels"], targets[b]["masks"], pc)
            i, j = linear_sum_assignment(C.cpu().numpy())
            idx.append((torch.as_tensor(i, dtype=torch.int64), torch.as_tensor(j, dtype=torch.int64)))
    return idx


# =====================================================================================================
# a15-a16  SetCriterion / SetCriterion_SS             modeling/criterion.py:121-287, criterion_ss.py:238-289
# =====================================================================================================

def uncertain_point_coords(src_masks, num_points, oversample_ratio, importance_sample_ratio, rand=torch.rand, topk_mask=None,
                           record=None):
    """detectron2 get_uncertain_point_coords_with_randomness with uncertainty = -|logit| (criterion.py:70-84,159-165).
    topk_mask: bool [n, oversampled points] - test hook: the frozen top-k SETS of the reference (golden `*/topk_bits`) instead
    of this function's own selection (the random stream is consumed identically; the order inside a set only changes the
    summation order of the point losses)."""
    n = src_masks.shape[0]
    ns = int(num_points * oversample_ratio)
    pc = rand(n, ns, 2)
    nu = int(importance_sample_ratio * num_points)
    nr = num_points - nu
    if topk_mask is not None:
        assert topk_mask.shape == (n, ns) and bool((topk_mask.sum(1) == nu).all())
        pc = pc[topk_mask].view(n, nu, 2)
    else:
        logits = point_sample(src_masks, pc)
        unc = -logits.abs()
        idx = torch.topk(unc[:, 0, :], k=nu, dim=1)[1]
        if record is not None:  # test hook: this run's own top-k SETS, in the form `topk_mask` accepts
            record.append(torch.zeros(n, ns, dtype=torch.bool).scatter_(1, idx, True))
        idx = idx + ns * torch.arange(n, dtype=torch.long)[:, None]
        pc = pc.view(-1, 2)[idx.view(-1)].view(n, nu, 2)
    if nr > 0:
        pc = torch.cat([pc, rand(n, nr, 2)], 1)
    return pc


def loss_labels(pred_logits, targets, indices, num_classes, eos_coef=0.1):
    """criterion.py:121-135: weighted CE over all queries, class weights [1,...,1,eos]."""
    B, Q, _ = pred_logits.shape
    tc = torch.full((B, Q), num_classes, dtype=torch.int64)
    for b, (src, tgt) in enumerate(indices):
        tc[b, src] = targets[b]["labels"][tgt]
    w = torch.ones(num_classes + 1)
    w[-1] = eos_coef
    return F.cross_entropy(pred_logits.float().transpose(1, 2), tc, w)


def loss_masks(pred_masks, targets, indices, num_masks, num_points=12544, oversample=3.0, importance=0.75,
               rand=torch.rand, topk_mask=None, record=None):
    """criterion.py:137-186 -> (loss_mask, loss_dice)"""
    src = torch.cat([pred_masks[b, s] for b, (s, _) in enumerate(indices)])  # [Nm,h,w]
    tgt = torch.cat([targets[b]["masks"][t] for b, (_, t) in enumerate(indices)]).to(src.dtype)
    src, tgt = src[:, None], tgt[:, None]
    with torch.no_grad():
        pc = uncertain_point_coords(src, num_points, oversample, importance, rand, topk_mask, record)
        labels = point_sample(tgt, pc).squeeze(1)
    logits = point_sample(src, pc).squeeze(1)
    l_mask = F.binary_cross_entropy_with_logits(logits, labels, reduction="none").mean(1).sum() / num_masks  # :44-62
    s = logits.sigmoid()
    l_dice = (1 - (2 * (s * labels).sum(-1) + 1) / (s.sum(-1) + labels.sum(-1) + 1)).sum() / num_masks  # :19-38
    return l_mask, l_dice


def similarity_loss(middle, n_frame=5):
    """criterion.py:208-231: c_f = 1-cos(m_f, m_{f+1}); sum_f c_f*exp(-c_f); /clips /(n_frame-1)."""
    bt, q, hw = middle.shape
    bs = bt // n_frame
    m = middle.reshape(bs, n_frame, q * hw)
    total = 0
    for f in range(n_frame - 1):
        x1, x2 = m[:, f], m[:, f + 1]
        cos = (x1 * x2).sum(1) / torch.sqrt(((x1 * x1).sum(1) + 1e-12) * ((x2 * x2).sum(1) + 1e-12))
        c = 1 - cos  # CosineEmbeddingLoss, target = +1
        total = total + c * torch.exp(-c)
    return total.sum() / bs / (n_frame - 1)


def set_criterion(outputs, targets, num_classes=2, gt_frame_index=None, world_size=1, rand=torch.rand,
                  num_points=12544, n_frame=5, frozen=None, record=None):
    """SetCriterion.forward (criterion.py:233-287).  `gt_frame_index`: None -> S4 rule (frames 0,5,10,... when
    len(outputs) != len(targets), :241-254); a LongTensor -> AVSS rule (criterion_ss.py:246-257).
    Returns the 39 un-weighted losses keyed like the reference.
    frozen: test hook - {"match_src", "match_tgt": int64 [10, Nm], "topk": bool [10, Nm, oversampled points]}: the reference's
    own discrete choices (golden `*/match_all_*`, `*/topk_bits`) replace the matcher's / the importance sampling's, so that a
    gradient comparison is not at the mercy of a near-tie falling the other way.
    record: test hook - a dict that receives THIS run's choices in the same form ("match_src", "match_tgt", "topk"), for
    injection into the implementation under test at sizes the golden fixtures do not cover (bs = 8)."""
    rec_src, rec_tgt, rec_topk = [], [], []

    def select(t):
        if gt_frame_index is not None:
            return t.index_select(0, gt_frame_index)
        if t.shape[0] != len(targets):
            return t.index_select(0, torch.arange(0, t.shape[0], 5))
        return t
    layers = [(select(outputs["pred_logits"]), select(outputs["pred_masks"]))]
    layers += [(select(a["pred_logits"]), select(a["pred_masks"])) for a in outputs["aux_outputs"]]
    num_masks = max(float(sum(len(t["labels"]) for t in targets)) / world_size, 1.0)  # :261-265
    losses = {}
    for li, (lg, mk) in enumerate(layers):  # final first, then aux 0..8  (:259, :268-277)
        ind = hungarian_matcher(lg, mk, targets, num_points, rand)
        topk_mask = None
        if frozen is not None:
            src, tgt = frozen["match_src"][li], frozen["match_tgt"][li]
            sizes = [len(t["labels"]) for t in targets]
            ind = list(zip(torch.split(torch.as_tensor(src, dtype=torch.int64), sizes),
                           torch.split(torch.as_tensor(tgt, dtype=torch.int64), sizes)))
            topk_mask = frozen["topk"][li]
        sfx = "" if li == 0 else f"_{li - 1}"
        losses["loss_ce" + sfx] = loss_labels(lg, targets, ind, num_classes)
        rec_src.append(torch.cat([i for i, _ in ind]))
        rec_tgt.append(torch.cat([j for _, j in ind]))
        lm, ld = loss_masks(mk, targets, ind, num_masks, num_points, rand=rand, topk_mask=topk_mask,
                            record=rec_topk if record is not None else None)
        losses["loss_mask" + sfx] = lm
        losses["loss_dice" + sfx] = ld
    for i, mid in enumerate(outputs["middles_attn_mask"]):  # :282-286
        losses[f"loss_cosine_{i}"] = similarity_loss(mid, n_frame)
    if record is not None:
        record.update(match_src=torch.stack(rec_src), match_tgt=torch.stack(rec_tgt))
        if rec_topk:
            record["topk"] = torch.stack(rec_topk)
    return losses


def loss_weights(dec_layers=10, w_ce=2.0, w_mask=5.0, w_dice=5.0, w_cos=10.0):
    """weight_dict of maskformer_model.py:200-211 (incl. the unused plain 'loss_cosine')."""
    base = {"loss_ce": w_ce, "loss_mask": w_mask, "loss_dice": w_dice, "loss_cosine": w_cos}
    wd = dict(base)
    for i in range(dec_layers - 1):
        wd.update({f"{k}_{i}": v for k, v in base.items()})
    return wd


# =====================================================================================================
# a17  inference tail                                  models/maskformer_model.py:393-441, 460-471
# =====================================================================================================

def semantic_inference(pred_logits, pred_masks, out_size, vid_flag=None):
    up = F.interpolate(pred_masks, size=out_size, mode="bilinear", align_corners=False)  # :397-402
    res = []
    for i, (c, m) in enumerate(zip(pred_logits, up)):
        r = torch.einsum("qc,qhw->chw", F.softmax(c, dim=-1)[..., :-1], m.sigmoid())  # :460-464
        if vid_flag is not None:
            r = r * vid_flag[i]  # :466-471
        res.append(r)
    return torch.stack(res)


# =====================================================================================================
# Host-PyTorch backbones (OUTSIDE the hot path; "parity unpinned": detectron2's ResNet is not installed).
# Restated from the public detectron2 0.6 ResNet-50 (BasicStem + Bottleneck, FrozenBN, STRIDE_IN_1X1=False,
# cfg at configs/avs_s4/R50-AVSS4-SemanticSegmentation.yaml:2-23) and torchvggish VGG (vggish.py:9-27, 89-100).
# =====================================================================================================

def _frozen_bn(P, pre, x, eps=1e-5):
    scale = P[pre + "weight"] * (P[pre + "running_var"] + eps).rsqrt()
    bias = P[pre + "bias"] - P[pre + "running_mean"] * scale
    return x * scale[None, :, None, None] + bias[None, :, None, None]


R50_STAGES = (("res2", 3, 64, 256, 1), ("res3", 4, 128, 512, 2), ("res4", 6, 256, 1024, 2), ("res5", 3, 512, 2048, 2))


def resnet50(P, pre, x):
    y = F.relu(_frozen_bn(P, pre + "stem.conv1.norm.", F.conv2d(x, P[pre + "stem.conv1.weight"], stride=2, padding=3)))
    y = F.max_pool2d(y, 3, 2, 1)
    feats = {}
    for name, nblk, mid, outc, stride in R50_STAGES:
        for b in range(nblk):
            p = f"{pre}{name}.{b}."
            s = stride if b == 0 else 1
            sc = y
            if p + "shortcut.weight" in P:
                sc = _frozen_bn(P, p + "shortcut.norm.", F.conv2d(y, P[p + "shortcut.weight"], stride=s))
            o = F.relu(_frozen_bn(P, p + "conv1.norm.", F.conv2d(y, P[p + "conv1.weight"])))
            o = F.relu(_frozen_bn(P, p + "conv2.norm.", F.conv2d(o, P[p + "conv2.weight"], stride=s, padding=1)))
            o = _frozen_bn(P, p + "conv3.norm.", F.conv2d(o, P[p + "conv3.weight"]))
            y = F.relu(o + sc)
        feats[name] = y
    return feats


def vggish(P, pre, x):
    """[N,1,96,64] log-mel -> [N,128] (vggish.py:18-27; PCA post-processing disabled by config)."""
    i = 0
    for v in (64, "M", 128, "M", 256, 256, "M", 512, 512, "M"):
        if v == "M":
            x = F.max_pool2d(x, 2, 2)
            i += 1
        else:
            x = F.relu(F.conv2d(x, P[f"{pre}features.{i}.weight"], P[f"{pre}features.{i}.bias"], padding=1))
            i += 2
    x = x.permute(0, 2, 3, 1).reshape(x.shape[0], -1)
    for j in (0, 2, 4):
        x = F.relu(F.linear(x, P[f"{pre}embeddings.{j}.weight"], P[f"{pre}embeddings.{j}.bias"]))
    return x


# PVTv2-B5 (round 6; pinned by tests/golden/pvt.npz, generated from the reference's own module: tests/golden/gen_golden_pvt.py)
# models/modeling/backbone/pvtv2.py:391-409: embed_dims 64 / 128 / 320 / 512, heads 1 / 2 / 5 / 8, depths 3 / 6 / 40 / 3,
# sr_ratios 8 / 4 / 2 / 1, mlp_ratio 4, qkv_bias, LayerNorm eps 1e-6 for the blocks' and stages' norms (norm_layer partial, :403)
# but the DEFAULT 1e-5 for the patch embeddings' and the spatial-reduction norms (plain nn.LayerNorm: :76, :203).
PVT_B5 = dict(dims=(64, 128, 320, 512), heads=(1, 2, 5, 8), depths=(3, 6, 40, 3), sr=(8, 4, 2, 1))


def _ln(P, pre, x, eps):
    return F.layer_norm(x, (x.shape[-1],), P[pre + "weight"], P[pre + "bias"], eps)


def pvt_attention(P, pre, x, H, W, heads, sr):
    """Attention.forward, pvtv2.py:104-132 (linear = False): queries from all N tokens, keys / values from the map reduced by a
    stride-sr convolution + LayerNorm; softmax(q k^T * d^-0.5) v; output projection."""
    B, N, C = x.shape
    d = C // heads
    q = F.linear(x, P[pre + "q.weight"], P[pre + "q.bias"]).reshape(B, N, heads, d).permute(0, 2, 1, 3)
    if sr > 1:
        x_ = x.permute(0, 2, 1).reshape(B, C, H, W)
        x_ = F.conv2d(x_, P[pre + "sr.weight"], P[pre + "sr.bias"], stride=sr).reshape(B, C, -1).permute(0, 2, 1)
        x_ = _ln(P, pre + "norm.", x_, 1e-5)
    else:
        x_ = x
    kv = F.linear(x_, P[pre + "kv.weight"], P[pre + "kv.bias"]).reshape(B, -1, 2, heads, d).permute(2, 0, 3, 1, 4)
    attn = ((q @ kv[0].transpose(-2, -1)) * d ** -0.5).softmax(dim=-1)
    y = (attn @ kv[1]).transpose(1, 2).reshape(B, N, C)
    return F.linear(y, P[pre + "proj.weight"], P[pre + "proj.bias"])


def pvt_mlp(P, pre, x, H, W):
    """Mlp.forward, pvtv2.py:48-57: fc1 -> 3x3 depth-wise convolution on the token map (DWConv, :377-386) -> GELU -> fc2"""
    B, N, _ = x.shape
    h = F.linear(x, P[pre + "fc1.weight"], P[pre + "fc1.bias"])
    C = h.shape[-1]
    h = F.conv2d(h.transpose(1, 2).reshape(B, C, H, W), P[pre + "dwconv.dwconv.weight"], P[pre + "dwconv.dwconv.bias"], padding=1, groups=C)
    h = F.gelu(h.flatten(2).transpose(1, 2))
    return F.linear(h, P[pre + "fc2.weight"], P[pre + "fc2.bias"])


def pvtv2_b5(P, pre, x, drop_path=None):
    """PyramidVisionTransformerV2.forward_features, pvtv2.py:343-362 -> {"res2".."res5"} NCHW.  Evaluation semantics (stochastic
    depth off) unless `drop_path` gives, per block, a pair of per-sample multipliers [B] (mask / keep probability) for its two
    residual branches (timm DropPath, pvtv2.py:162-175)."""
    outs = {}
    B = x.shape[0]
    k = 0
    for i, (dim, heads, depth, sr) in enumerate(zip(PVT_B5["dims"], PVT_B5["heads"], PVT_B5["depths"], PVT_B5["sr"])):
        pe = f"{pre}patch_embed{i + 1}."
        ks = 7 if i == 0 else 3
        x = F.conv2d(x, P[pe + "proj.weight"], P[pe + "proj.bias"], stride=4 if i == 0 else 2, padding=ks // 2)  # :214-219
        H, W = x.shape[-2:]
        x = _ln(P, pe + "norm.", x.flatten(2).transpose(1, 2), 1e-5)
        for j in range(depth):
            bp = f"{pre}block{i + 1}.{j}."
            a = pvt_attention(P, bp + "attn.", _ln(P, bp + "norm1.", x, 1e-6), H, W, heads, sr)
            m1 = 1.0 if drop_path is None else drop_path[k][0].view(B, 1, 1)
            x = x + m1 * a
            m = pvt_mlp(P, bp + "mlp.", _ln(P, bp + "norm2.", x, 1e-6), H, W)
            m2 = 1.0 if drop_path is None else drop_path[k][1].view(B, 1, 1)
            x = x + m2 * m
            k += 1
        x = _ln(P, f"{pre}norm{i + 1}.", x, 1e-6)
        x = x.reshape(B, H, W, -1).permute(0, 3, 1, 2)
        outs[f"res{i + 2}"] = x
    return outs


# MaskFormer.forward                                    models/maskformer_model.py:274-441
PIXEL_MEAN = (123.675, 116.280, 103.530)
PIXEL_STD = (58.395, 57.120, 57.375)


def maskformer_forward(P, batched_inputs, num_classes=2, training=True, rand=torch.rand, world_size=1, record=None, backbone="r50",
                       avss=False, attn_override=None, frozen=None):
    """Full model step on CPU: normalise, VGGish (no grad), dual R50 / dual PVTv2-B5 (`backbone`), SEM mix, head, then the weighted
    39-term loss (training) or the per-frame sem_seg maps (eval).  avss (round 6): the AVSS data path - the flag tensors of the
    clips are concatenated, the audio rows of the frames that exist are kept (maskformer_model.py:300-331) and the criterion
    selects the annotated frames (criterion_ss.py:246-257); the inference tail multiplies by the frame flag (:466-471).
    attn_override / frozen: test hooks handed to the decoder / criterion (the reference's or another run's discrete choices).
    record: test hook - a dict that receives this run's discrete choices: "attn_masks" (list of 9 bool [BT,Q,hw], the form
    `transformer_decoder(attn_override=...)` accepts), the criterion's "match_src" / "match_tgt" / "topk", and the 10 heads'
    "pred_masks" [BT,Q,H/4,W/4] / "pred_logits" [BT,Q,K+1] (transformer_decoder.py:481-509)."""
    mean = torch.tensor(PIXEL_MEAN).view(1, 3, 1, 1)
    std = torch.tensor(PIXEL_STD).view(1, 3, 1, 1)
    images = torch.cat([b["images"] for b in batched_inputs]).float()
    pre_masks = torch.cat([b["pre_masks"] for b in batched_inputs]).float()
    mel = torch.cat([b["audio_log_mel"] for b in batched_inputs])
    images = (images - mean) / std  # :324   (224 is a multiple of SIZE_DIVISIBILITY=32: ImageList pads nothing)
    pre_masks = (pre_masks - mean) / std  # :338
    with torch.no_grad():
        audio = vggish(P, "audio_backbone.", mel).unsqueeze(1)  # :327-329
    vid_flag = gt_flag = None
    if avss:
        vid_flag = torch.cat([b["vid_temporal_mask_flag"] for b in batched_inputs])  # :300-312
        gt_flag = torch.cat([b["gt_temporal_mask_flag"] for b in batched_inputs])
        audio = audio[vid_flag.bool()]  # :330-331
    net = resnet50 if backbone == "r50" else pvtv2_b5
    feats = net(P, "backbone.", images)  # :333
    pfeats = net(P, "pre_sam_backbone.", pre_masks)  # :341
    feats = sem_mix(P, "scale_factor_module.", feats, pfeats)  # :345-352
    out = head_forward(P, "sem_seg_head.", feats, audio, attn_override=attn_override)  # :363
    if not training:
        return semantic_inference(out["pred_logits"], out["pred_masks"], images.shape[-2:], vid_flag=vid_flag)
    targets = []
    for b in batched_inputs:  # prepare_targets :443-458 (no padding needed at 224)
        for inst in b["instances"]:
            targets.append({"labels": inst["gt_classes"], "masks": inst["gt_masks"]})
    gt_index = torch.where(gt_flag == 1)[0] if avss else None  # criterion_ss.py:246
    losses = set_criterion(out, targets, num_classes, gt_index, world_size, rand, record=record, frozen=frozen)
    if record is not None:
        record["attn_masks"] = [a[::8].clone() for a in out["attn_masks"][:9]]  # one of the 8 identical head replicas
        # the north-star's compared quantity: mask / class logits of all 10 prediction heads (9 auxiliary + the final one)
        record["pred_masks"] = [a["pred_masks"].detach() for a in out["aux_outputs"]] + [out["pred_masks"].detach()]
        record["pred_logits"] = [a["pred_logits"].detach() for a in out["aux_outputs"]] + [out["pred_logits"].detach()]
    wd = loss_weights()
    return {k: v * wd[k] for k, v in losses.items()}  # :384-391


# ---------------------------------------------------------------------------------------- evaluator metric
def mask_iou(pred, target, eps=1e-7):
    """models/evaluation/sem_seg_evaluation.py:66-92: pred [N,H,W] probabilities (thresholded at 0.5), target [N,H,W] 0/1.
    A frame with EMPTY ground truth scores the agreement of the backgrounds over all pixels (:83-89)."""
    n = pred.shape[0]
    p = (pred > 0.5).to(target.dtype)
    pixels = pred.shape[-1] * pred.shape[-2]
    empty = target.sum((1, 2)) == 0
    inter = (p * target).sum((1, 2))
    union = torch.max(p, target).sum((1, 2))
    inter = torch.where(empty, ((1 - target) * (1 - p)).sum((1, 2)), inter)
    union = torch.where(empty, torch.full_like(union, float(pixels)), union)
    return (inter / (union + eps)).sum() / n


def eval_fmeasure(pred, gt, pr_num=255):
    """:95-137: maximum over 255 thresholds of the F-beta (beta^2 = 0.3) curve averaged over the frames with non-empty ground
    truth; NaN entries of a frame's curve (0/0) count as 0."""
    beta2 = 0.3
    th = torch.linspace(0, 1 - 1e-10, pr_num)
    curves = []
    for i in range(pred.shape[0]):
        if gt[i].mean() == 0.0:
            continue
        y = (pred[i][None] >= th[:, None, None]).float()
        tp = (y * gt[i][None]).sum((1, 2))
        prec, rec = tp / (y.sum((1, 2)) + 1e-20), tp / (gt[i].sum() + 1e-20)
        f = (1 + beta2) * prec * rec / (beta2 * prec + rec)
        curves.append(torch.nan_to_num(f, nan=0.0))
    if not curves:
        return 0.0
    return float(torch.stack(curves).mean(0).max())


def s4_clip_metrics(sem_seg, gts):
    """SemSegEvaluator.process :219-245 for one batch: sem_seg [N,K,H,W] (the model's eval output per frame) -> softmax over K
    (a SECOND softmax: the inference tail already mixed class probabilities, Appendix A) -> channel 1 -> (mIoU, F-score)."""
    probs = F.softmax(sem_seg, dim=1)[:, 1]
    return float(mask_iou(probs, gts)), eval_fmeasure(probs, gts)


# ---------------------------------------------------------------------------------------- AVSS evaluator metric
def avss_batch_miou_fscore(output, target, nclass, T=10, beta2=0.3):
    """models/evaluation/sem_seg_evaluation_ss.py:66-104 `_batch_miou_fscore`, frame by frame with torch.histc like the
    reference: output [BF,C,H,W] scores, target [BF,H,W] class ids -> (sum of per-frame IoU per class, sum of per-frame F
    per class, number of frames in which a class has a non-empty union, per-frame mean IoU over the classes with IoU != 0)."""
    predict = (torch.argmax(output, 1) + 1).float()
    target = target.float() + 1
    predict = predict * (target > 0).float()
    inter = predict * (predict == target).float()
    ious, fscores, cls_count, vid = torch.zeros(nclass), torch.zeros(nclass), torch.zeros(nclass), []
    for i in range(target.shape[0]):
        a_i = torch.histc(inter[i], bins=nclass, min=1, max=nclass)
        a_p = torch.histc(predict[i], bins=nclass, min=1, max=nclass)
        a_l = torch.histc(target[i], bins=nclass, min=1, max=nclass)
        a_u = a_p + a_l - a_i
        iou = a_i / (2.220446049250313e-16 + a_u)
        ious += iou
        cls_count[a_u != 0] += 1
        prec, rec = a_i / a_p, a_i / a_l
        f = (1 + beta2) * prec * rec / (beta2 * prec + rec)
        f[torch.isnan(f)] = 0.0
        fscores += f
        vid.append(iou.sum() / (iou != 0).float().sum())
    return ious, fscores, cls_count, vid


def avss_calc_color_miou_fscore(pred, target, T=10):
    """sem_seg_evaluation_ss.py:107-118: softmax over the class axis first (does not move the arg-max)"""
    return avss_batch_miou_fscore(torch.softmax(pred, dim=1), target, pred.shape[1], T)


def avss_evaluate(batches):
    """SemSegEvaluator_SS.evaluate on one process (:254-266).  batches: list of (miou, fscore, cls_count) from `process`
    -> {"mIoU", "f_score"} rounded to 4 digits like the reference, + the no-background means."""
    n = len(batches)
    miou_pc = sum(b[0] for b in batches) / n
    f_pc = sum(b[1] for b in batches) / n
    cls_pc = sum(b[2] for b in batches) / n
    miou_pc = miou_pc / cls_pc
    miou_pc[torch.isnan(miou_pc)] = 0
    f_pc = f_pc / cls_pc
    f_pc[torch.isnan(f_pc)] = 0
    return {"mIoU": round(miou_pc.mean().item(), 4), "f_score": round(f_pc.mean().item(), 4),
            "mIoU_noBg": miou_pc[:-1].mean().item(), "f_score_noBg": f_pc[:-1].mean().item()}
