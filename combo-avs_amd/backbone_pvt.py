"""Host-PyTorch PVTv2-B5 backbone (outside the HIP hot path, BASELINE.json north_star; SURVEY 8(b) registry surface and
8(f) rank 2): `BACKBONE_REGISTRY["build_pvtv2_b5_backbone"]`, the visual encoder of the reference's headline COMBO-PVT
configs (models/modeling/backbone/pvtv2.py:236-409).

Written from the published architecture (Pyramid Vision Transformer v2: four stages of overlapping patch embedding +
transformer blocks with spatial-reduction attention and a depth-wise-conv MLP) with the reference's hyper-parameters
(pvtv2.py:391-409: dims 64/128/320/512, heads 1/2/5/8, depths 3/6/40/3, sr 8/4/2/1, qkv bias, LayerNorm eps 1e-6,
stochastic depth 0.1 with the linear decay rule) and its parameter names (`patch_embed{i}.proj|norm`,
`block{i}.{j}.{norm1,attn.{q,kv,sr,norm,proj},norm2,mlp.{fc1,dwconv.dwconv,fc2}}`, `norm{i}`), so reference
checkpoints load 1:1.

MI355X notes: tokens stay batch-first [B,N,C]; the spatial-reduction attention goes through
`F.scaled_dot_product_attention` (no [B,heads,N,N'] score tensor in HBM: at 512x512 stage 1 that tensor alone is
B x 16384 x 256 floats); the stage output is returned as an NCHW *view* of the token-major tensor (channels_last
memory), which is what the SEM mix and the pixel decoder consume; stochastic depth draws its per-sample mask on the
device (graph-capturable)."""
import math

import torch
from torch import nn
from torch.nn import functional as F

from .registry import BACKBONE_REGISTRY, ShapeSpec


def drop_path(x, p, training):
    """Stochastic depth per sample (timm DropPath semantics: keep with prob 1-p, rescale by 1/(1-p))."""
    if p == 0.0 or not training:
        return x
    keep = 1.0 - p
    mask = (torch.rand(x.shape[0], *([1] * (x.dim() - 1)), device=x.device, dtype=x.dtype) < keep).to(x.dtype)
    return x * (mask / keep)


# (round 6) the spatial-reduction convolution (kernel = stride = sr, no padding: pvtv2.py:76) as a GEMM over non-overlapping patches:
# [B * Ho * Wo, sr * sr * C] x [Cout, sr * sr * C]^T.  Two reasons: MIOpen's default (immediate-mode) bf16 solver for these layers
# is not run-to-run reproducible (profiles/r06_pvt_eager_vs_replay_states.txt - it was the source of the eager-vs-replay "states" of
# the PVT recipe).  An OPTION for reproducible runs without `torch.backends.cudnn.deterministic`: measured 105.9 vs 105.1 ms per
# `pvt_ms3_t10` step against MIOpen's find-mode kernels (same box, A B A B: profiles/r06_ab_pvt_sr_patch_gemm.txt), so the default
# stays F.conv2d.
SR_PATCH_GEMM = False
PRENORM = True  # False: the per-op formulation (the reference of tests/test_kernels_gpu.py::test_pvt_prenorm_path_matches_per_op_path)


class _NullCtx:
    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False


class DropPath(nn.Module):
    def __init__(self, p=0.0):
        super().__init__()
        self.p = float(p)

    def forward(self, x):
        return drop_path(x, self.p, self.training)

    def extra_repr(self):
        return f"p={self.p}"


class _CastAll(torch.autograd.Function):
    """bf16 copies of ALL linear / convolution parameters of the backbone as one autograd node (two multi-tensor launches
    per direction) instead of autocast's one cast kernel per parameter and step (~1 000 forward + ~900 backward launches
    for PVTv2-B5)."""

    @staticmethod
    def forward(ctx, dtype, queue, *params):
        ctx.queue = queue  # this application's deferred bias gradients (ops.colsum.DeferredColumnSums) or None
        out = [torch.empty_like(p, dtype=dtype) for p in params]
        if params[0].is_cuda and all(p.is_contiguous() for p in params):
            from .ops.foldcast import fold_cast  # csrc/foldcast.hip: tables of 56 tensors per launch
            fold_cast([p.detach() for p in params], out)
        else:
            torch._foreach_copy_(out, params)
        return tuple(out)

    @staticmethod
    def backward(ctx, *grads):
        if ctx.queue is not None:
            ctx.queue.flush()  # the bias gradients queued by _linear (ops/colsum.py): one grouped launch instead of ~300 pairs
        idx = [i for i, g in enumerate(grads) if g is not None]
        g32 = [torch.empty(grads[i].shape, dtype=torch.float32, device=grads[i].device) for i in idx]
        if idx and grads[idx[0]].is_cuda:
            from .ops.foldcast import fold_cast
            # the strided-convolution weight gradients arrive channels_last: one strided copy_ each (a .contiguous() for the
            # grouped cast would be a second kernel over the same tensor)
            dense = [j for j, i in enumerate(idx) if grads[i].is_contiguous()]
            for j, i in enumerate(idx):
                if not grads[i].is_contiguous():
                    g32[j].copy_(grads[i])
            if dense:
                fold_cast([grads[idx[j]] for j in dense], [g32[j] for j in dense])
        elif idx:
            torch._foreach_copy_(g32, [grads[i] for i in idx])
        out = [None] * len(grads)
        for i, g in zip(idx, g32):
            out[i] = g
        return (None, None) + tuple(out)


# Explicit-precision helpers.  `wts` maps id(parameter) -> its compute-dtype copy (from _CastAll) when the backbone runs
# in reduced precision; None = use the module's own parameters (fp32 / eval / CPU).  The dtype rules are autocast's:
# linear / convolution / attention in the compute dtype, LayerNorm in fp32, residual sums promote to fp32.
def _p(param, wts):
    return param if wts is None else wts.get(id(param), param)


# Bias gradients: autograd's grad_output.sum(...) splits these long reductions over workgroups behind a memset node, which a
# replayed hipGraph does not execute reliably on this stack (ops/colsum.py, tools/graph_reduce_repro.py) - the linears and
# convolutions below take their bias gradient from csrc/colsum.hip instead.
def _linear(x, mod, wts):
    w = _p(mod.weight, wts)
    x = x if x.dtype == w.dtype else x.to(w.dtype)
    if mod.bias is None:
        return F.linear(x, w)
    if x.is_cuda and torch.is_grad_enabled():
        from .ops.colsum import linear_bias
        # with the batched casts (wts) every bias gradient is consumed by _CastAll.backward, which flushes the queue first
        return linear_bias(x, w, _p(mod.bias, wts), queue=None if wts is None else wts.get("colsum_queue"))
    return F.linear(x, w, _p(mod.bias, wts))


def _conv(x, mod, wts):
    w = _p(mod.weight, wts)
    x = x if x.dtype == w.dtype else x.to(w.dtype)
    if mod.bias is not None and x.is_cuda and torch.is_grad_enabled():
        from .ops.colsum import add_channel_vector
        return add_channel_vector(F.conv2d(x, w, None, mod.stride, mod.padding, mod.dilation, mod.groups), _p(mod.bias, wts), 1)
    return F.conv2d(x, w, None if mod.bias is None else _p(mod.bias, wts), mod.stride, mod.padding, mod.dilation, mod.groups)


def _sr_patch_gemm(xs, w, B, H, W, C, sr):
    """conv2d(x, w, stride = kernel = sr) on token-major x [B, H * W, C] -> [B, Ho * Wo, Cout]: the non-overlapping sr x sr patches
    are the rows of a GEMM (one gather copy of x, one library GEMM; rows / columns beyond Ho * sr, Wo * sr are dropped as the
    convolution drops them)"""
    Ho, Wo = H // sr, W // sr
    x4 = xs.view(B, H, W, C)
    if Ho * sr != H or Wo * sr != W:
        x4 = x4[:, :Ho * sr, :Wo * sr]
    patches = x4.reshape(B, Ho, sr, Wo, sr, C).permute(0, 1, 3, 2, 4, 5).reshape(B * Ho * Wo, sr * sr * C)
    w2 = w.permute(0, 2, 3, 1).reshape(w.shape[0], sr * sr * C)  # [Cout, (ky, kx, c)]
    return F.linear(patches, w2).view(B, Ho * Wo, w.shape[0])


def _ln(x, mod, wts):
    if wts is None:
        return F.layer_norm(x, mod.normalized_shape, mod.weight, mod.bias, mod.eps)
    x = x.float()
    if x.is_cuda and torch.is_grad_enabled() and mod.weight.requires_grad:
        # every LayerNorm of the backbone is applied once per forward: its parameter gradients may be deferred into the
        # grouped launch (ops/layernorm.py), ~320 fewer reduction kernels per backbone backward
        from .ops.layernorm import _AddLayerNorm, _own_ok
        if _own_ok(x, x.shape[-1]):
            return _AddLayerNorm.apply(x, None, mod.weight, mod.bias, mod.eps, True)  # csrc/layernorm.hip
    return F.layer_norm(x, mod.normalized_shape, mod.weight, mod.bias, mod.eps)


def _init(m):
    """pvtv2.py:324-337: trunc-normal(0.02) linears, unit LayerNorm, fan-out normal convolutions."""
    if isinstance(m, nn.Linear):
        nn.init.trunc_normal_(m.weight, std=0.02)
        if m.bias is not None:
            nn.init.zeros_(m.bias)
    elif isinstance(m, nn.LayerNorm):
        nn.init.ones_(m.weight)
        nn.init.zeros_(m.bias)
    elif isinstance(m, nn.Conv2d):
        fan_out = m.kernel_size[0] * m.kernel_size[1] * m.out_channels // m.groups
        nn.init.normal_(m.weight, 0.0, math.sqrt(2.0 / fan_out))
        if m.bias is not None:
            nn.init.zeros_(m.bias)


class DWConv(nn.Module):
    """3x3 depth-wise convolution on the token grid (pvtv2.py:377-388)."""

    def __init__(self, dim):
        super().__init__()
        self.dwconv = nn.Conv2d(dim, dim, 3, 1, 1, bias=True, groups=dim)

    def forward(self, x, H, W):
        B, N, C = x.shape
        from .ops import dwconv
        xt = x.view(B, H, W, C)
        w, b = self.dwconv.weight, self.dwconv.bias
        if dwconv.usable(xt, w):  # bf16 tokens on the GPU: bandwidth-bound HIP kernels (csrc/dwconv.hip), fp32 weights
            return dwconv.dwconv3x3(xt, w, b).view(B, N, C)
        y = self.dwconv(xt.permute(0, 3, 1, 2))  # NCHW view of token-major memory (channels_last)
        return y.permute(0, 2, 3, 1).reshape(B, N, C)


class Mlp(nn.Module):
    """fc1 -> depth-wise conv -> GELU -> fc2 (pvtv2.py:17-57, linear=False)."""

    def __init__(self, dim, hidden, drop=0.0):
        super().__init__()
        self.fc1 = nn.Linear(dim, hidden)
        self.dwconv = DWConv(hidden)
        self.act = nn.GELU()
        self.fc2 = nn.Linear(hidden, dim)
        self.drop = nn.Dropout(drop)

    def forward(self, x, H, W, wts=None):
        x = self.drop(self.act(self.dwconv(_linear(x, self.fc1, wts), H, W)))
        return self.drop(_linear(x, self.fc2, wts))


class Attention(nn.Module):
    """Spatial-reduction attention (pvtv2.py:60-132, linear=False): keys/values come from the token grid reduced by a
    strided sr x sr convolution + LayerNorm."""

    def __init__(self, dim, num_heads, qkv_bias, attn_drop=0.0, proj_drop=0.0, sr_ratio=1):
        super().__init__()
        assert dim % num_heads == 0, f"dim {dim} should be divided by num_heads {num_heads}."
        self.dim, self.num_heads, self.sr_ratio = dim, num_heads, sr_ratio
        self.scale = (dim // num_heads) ** -0.5
        self.q = nn.Linear(dim, dim, bias=qkv_bias)
        self.kv = nn.Linear(dim, dim * 2, bias=qkv_bias)
        self.attn_drop = nn.Dropout(attn_drop)
        self.proj = nn.Linear(dim, dim)
        self.proj_drop = nn.Dropout(proj_drop)
        if sr_ratio > 1:
            self.sr = nn.Conv2d(dim, dim, kernel_size=sr_ratio, stride=sr_ratio)
            self.norm = nn.LayerNorm(dim)

    def forward(self, x, H, W, wts=None, x_kv=None):
        """x_kv: an alias of x for the key / value path (ops/prenorm.py fan-out: the two consumers' gradients then meet inside the
        LayerNorm's backward kernel instead of in an accumulation kernel)"""
        B, N, C = x.shape
        h, d = self.num_heads, C // self.num_heads
        q = _linear(x, self.q, wts).view(B, N, h, d).transpose(1, 2)
        xs = x if x_kv is None else x_kv
        if self.sr_ratio > 1:
            w_sr = _p(self.sr.weight, wts)
            if wts is not None and PRENORM and xs.is_cuda and xs.dtype == w_sr.dtype == torch.bfloat16 and torch.is_grad_enabled():
                # reduced-precision training path: the convolution without its bias, then bias + LayerNorm -> bf16 in one pass
                from .ops.prenorm import bias_ln
                if SR_PATCH_GEMM:
                    x_ = _sr_patch_gemm(xs, w_sr, B, H, W, C, self.sr_ratio)  # [B, Ho * Wo, C], token-major
                else:
                    x_ = F.conv2d(xs.view(B, H, W, C).permute(0, 3, 1, 2), w_sr, None, self.sr.stride).flatten(2).transpose(1, 2)
                x_ = bias_ln(x_, _p(self.sr.bias, wts), self.norm)
            else:
                x_ = _conv(xs.view(B, H, W, C).permute(0, 3, 1, 2), self.sr, wts)  # [B,C,H/sr,W/sr]
                x_ = _ln(x_.flatten(2).transpose(1, 2), self.norm, wts)
        else:
            x_ = xs
        kv = _linear(x_, self.kv, wts)
        drop = self.attn_drop.p if self.training else 0.0
        if drop == 0.0:
            from .ops import sra
            q_lin = q.transpose(1, 2).reshape(B, N, C)  # (a view of the projection's output: q is its [B, h, N, d] view)
            if sra.usable(q_lin, kv, h):
                # own kernels (csrc/sra_attention.hip): few keys, head dimension 64 - the projections' outputs are read where they
                # lie, the result is written in the [B, N, C] layout the output projection consumes
                return self.proj_drop(_linear(sra.sra_attention(q_lin, kv, h, self.scale), self.proj, wts))
        k, v = kv.view(B, -1, 2, h, d).unbind(2)          # (unbind: ONE stack in the backward pass, not two zero-filled select
        k, v = k.transpose(1, 2), v.transpose(1, 2)      # gradients and their sum)
        o = F.scaled_dot_product_attention(q, k, v, dropout_p=drop, scale=self.scale)
        return self.proj_drop(_linear(o.transpose(1, 2).reshape(B, N, C), self.proj, wts))


class Block(nn.Module):
    def __init__(self, dim, num_heads, mlp_ratio, qkv_bias, drop, attn_drop, drop_path_p, norm_eps, sr_ratio):
        super().__init__()
        self.norm1 = nn.LayerNorm(dim, eps=norm_eps)
        self.attn = Attention(dim, num_heads, qkv_bias, attn_drop, drop, sr_ratio)
        self.drop_path = DropPath(drop_path_p) if drop_path_p > 0.0 else nn.Identity()
        self.norm2 = nn.LayerNorm(dim, eps=norm_eps)
        self.mlp = Mlp(dim, int(dim * mlp_ratio), drop)

    def forward(self, x, H, W, wts=None):
        x = x + self.drop_path(self.attn(_ln(x, self.norm1, wts), H, W, wts))
        return x + self.drop_path(self.mlp(_ln(x, self.norm2, wts), H, W, wts))


class OverlapPatchEmbed(nn.Module):
    """Overlapping patch embedding: strided convolution (kernel > stride) + LayerNorm (pvtv2.py:193-233; note the norm
    here is a default-eps LayerNorm, `:210`)."""

    def __init__(self, patch_size, stride, in_chans, embed_dim):
        super().__init__()
        assert patch_size > stride, "Set larger patch_size than stride"
        self.proj = nn.Conv2d(in_chans, embed_dim, kernel_size=patch_size, stride=stride, padding=patch_size // 2)
        self.norm = nn.LayerNorm(embed_dim)

    def forward(self, x, wts=None):
        x = _conv(x, self.proj, wts)
        H, W = x.shape[-2:]
        return _ln(x.flatten(2).transpose(1, 2), self.norm, wts), H, W


class PyramidVisionTransformerV2(nn.Module):
    # two instances on two HIP streams (meta_arch.MaskFormer.parallel_backbones): NOT safe - the captured `pvt_ms3_t10` step with the two
    # encoders side by side never finished its first replay (tools/job_pvt_parallel_probe.sh)
    concurrent_safe = False

    def __init__(self, in_chans=3, embed_dims=(64, 128, 256, 512), num_heads=(1, 2, 4, 8), mlp_ratios=(4, 4, 4, 4),
                 qkv_bias=False, drop_rate=0.0, attn_drop_rate=0.0, drop_path_rate=0.0, norm_eps=1e-5,
                 depths=(3, 4, 6, 3), sr_ratios=(8, 4, 2, 1), num_stages=4, out_features=None):
        super().__init__()
        self.depths, self.num_stages = list(depths), num_stages
        self._out_features = list(out_features or [f"res{i + 2}" for i in range(num_stages)])
        total = sum(depths)
        dpr = [drop_path_rate * i / max(total - 1, 1) for i in range(total)]  # linear stochastic-depth decay (:262)
        cur = 0
        self._out_feature_strides, self._out_feature_channels = {}, {}
        for i in range(num_stages):
            setattr(self, f"patch_embed{i + 1}", OverlapPatchEmbed(7 if i == 0 else 3, 4 if i == 0 else 2,
                                                                   in_chans if i == 0 else embed_dims[i - 1], embed_dims[i]))
            setattr(self, f"block{i + 1}", nn.ModuleList([
                Block(embed_dims[i], num_heads[i], mlp_ratios[i], qkv_bias, drop_rate, attn_drop_rate, dpr[cur + j], norm_eps,
                      sr_ratios[i]) for j in range(depths[i])]))
            setattr(self, f"norm{i + 1}", nn.LayerNorm(embed_dims[i], eps=norm_eps))
            cur += depths[i]
            stage = f"res{i + 2}"
            if stage in self._out_features:
                self._out_feature_strides[stage] = 4 if i == 0 else 2 ** (i + 2)
                self._out_feature_channels[stage] = embed_dims[i]
        self.size_divisibility = 0
        self.apply(_init)

    def freeze_patch_emb(self):
        self.patch_embed1.requires_grad_(False)

    def no_weight_decay(self):
        return {"pos_embed1", "pos_embed2", "pos_embed3", "pos_embed4", "cls_token"}

    def _cast_params(self):
        """every parameter autocast would cast per use: linear / convolution weights and biases, except the depth-wise
        convolutions (csrc/dwconv.hip reads fp32 weights)"""
        ps = []
        for name, mod in self.named_modules():
            if isinstance(mod, (nn.Linear, nn.Conv2d)) and not name.endswith("dwconv.dwconv"):
                ps += [p for p in (mod.weight, mod.bias) if p is not None]
        return ps

    def forward(self, x):
        wts = None
        if x.is_cuda and torch.is_autocast_enabled("cuda"):
            dtype = torch.get_autocast_dtype("cuda")
            params = self._cast_params()
            queue = None
            if torch.is_grad_enabled():
                from .ops.colsum import DeferredColumnSums
                queue = DeferredColumnSums()  # one per application: flushed by THIS application's _CastAll.backward
            wts = {id(p): c for p, c in zip(params, _CastAll.apply(dtype, queue, *params))}
            wts["colsum_queue"] = queue
        with torch.autocast("cuda", enabled=False) if wts is not None else _NullCtx():
            return self._forward(x, wts)

    def _drop_path_scales(self, B, device):
        """stochastic-depth multipliers (mask / keep) of ALL residual branches of one forward pass, [2 * blocks, B] fp32, from one
        random draw (the per-branch formulation costs three tiny launches per branch, ~600 per step for two PVTv2-B5)"""
        ps = [blk.drop_path.p if isinstance(blk.drop_path, DropPath) else 0.0
              for i in range(self.num_stages) for blk in getattr(self, f"block{i + 1}")]
        if not self.training or not any(ps):
            return None
        key = str(device)
        cache = self.__dict__.setdefault("_keep_cache", {})
        if key not in cache:
            cache[key] = (1.0 - torch.tensor(ps, dtype=torch.float32, device=device)).repeat_interleave(2)[:, None]
        keep = cache[key]
        return (torch.rand(keep.shape[0], B, device=device) < keep).to(torch.float32) / keep

    def _forward_prenorm(self, x, wts):
        """the reduced-precision training path: every LayerNorm of the blocks runs as ONE pre-norm residual step
        (ops/prenorm.py: residual add with the stochastic-depth multiplier + LayerNorm + cast to the compute dtype)"""
        from .ops.prenorm import prenorm
        B = x.shape[0]
        outs = {}
        scales = self._drop_path_scales(B, x.device)
        k = 0
        for i in range(self.num_stages):
            x, H, W = getattr(self, f"patch_embed{i + 1}")(x, wts)
            stream, branch, s = x.contiguous(), None, None
            for blk in getattr(self, f"block{i + 1}"):
                if branch is None:
                    y, y_kv = prenorm(stream, None, None, blk.norm1, fanout=2)
                else:
                    stream, y, y_kv = prenorm(stream, branch, s, blk.norm1, fanout=2)
                branch, s = blk.attn(y, H, W, wts, x_kv=y_kv), (None if scales is None else scales[k])
                stream, y = prenorm(stream, branch, s, blk.norm2)
                branch, s = blk.mlp(y, H, W, wts), (None if scales is None else scales[k + 1])
                k += 2
            _, x = prenorm(stream, branch, s, getattr(self, f"norm{i + 1}"), out_fp32=True)
            x = x.view(B, H, W, -1).permute(0, 3, 1, 2)
            stage = f"res{i + 2}"
            if stage in self._out_features:
                outs[stage] = x
        return outs

    def _forward(self, x, wts):
        if wts is not None and x.is_cuda and torch.is_grad_enabled() and PRENORM:
            return self._forward_prenorm(x, wts)
        B = x.shape[0]
        outs = {}
        for i in range(self.num_stages):
            x, H, W = getattr(self, f"patch_embed{i + 1}")(x, wts)
            for blk in getattr(self, f"block{i + 1}"):
                x = blk(x, H, W, wts)
            x = _ln(x, getattr(self, f"norm{i + 1}"), wts)
            x = x.view(B, H, W, -1).permute(0, 3, 1, 2)  # NCHW view, channels_last memory (the reference copies to NCHW, :359)
            stage = f"res{i + 2}"
            if stage in self._out_features:
                outs[stage] = x
        return outs

    def output_shape(self):
        return {n: ShapeSpec(channels=self._out_feature_channels[n], stride=self._out_feature_strides[n]) for n in self._out_features}


@BACKBONE_REGISTRY.register()
def build_pvtv2_b5_backbone(cfg, input_shape=None):
    """PVTv2-B5 (pvtv2.py:391-409)."""
    return PyramidVisionTransformerV2(
        embed_dims=(64, 128, 320, 512), num_heads=(1, 2, 5, 8), mlp_ratios=(4, 4, 4, 4), qkv_bias=True, norm_eps=1e-6,
        depths=(3, 6, 40, 3), sr_ratios=(8, 4, 2, 1), drop_rate=0.0, drop_path_rate=0.1,
        out_features=cfg.MODEL.PVT.OUT_FEATURES)
