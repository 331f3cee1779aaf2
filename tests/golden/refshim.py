"""Import harness for the UPSTREAM reference's hot-path modules (build container only).

This file is *our own* test tooling.  It does not contain reference code: it installs small
stand-ins for the third-party names the reference's hot-path files import but which are not
installed here (detectron2==0.6, fvcore, timm, torchvision -- SURVEY.md §8(c)), registers the
reference's top-level packages as bare namespace modules (so `models/__init__.py`, which pulls
data/eval/d2 wholesale, is never executed) and then lets Python import the leaf modules from
`/root/reference` at run time.  Used only by `gen_golden.py` to produce the committed fixtures.
It refuses to run when `/root/reference` is absent (e.g. on the GPU box).

Semantics of the stand-ins follow the public detectron2 0.6 / fvcore / timm behaviour:
  * `detectron2.layers.Conv2d`      : nn.Conv2d + optional `norm` + optional `activation`
  * `detectron2.layers.get_norm`    : "GN" -> GroupNorm(32, C); "" -> None
  * `fvcore...c2_xavier_fill`       : kaiming_uniform_(a=1) on weight, bias = 0
  * `point_sample`                  : grid_sample(x, 2*coords-1) (3-D coords get a dummy axis)
  * `get_uncertain_point_coords_with_randomness` : PointRend importance sampling
  * `timm DropPath`                 : identity at p=0 (only p=0 is used, fuse_helper.py:287)
"""
import os
import sys
import types

import torch
import torch.nn.functional as F
from torch import nn

REF_ROOT = "/root/reference"


def _mod(name, **attrs):
    m = types.ModuleType(name)
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    # attach to parent
    if "." in name:
        parent, child = name.rsplit(".", 1)
        if parent in sys.modules:
            setattr(sys.modules[parent], child, m)
    return m


class _Registry:
    def __init__(self, name):
        self._name = name
        self._map = {}

    def register(self, obj=None):
        if obj is None:
            def deco(o):
                self._map[o.__name__] = o
                return o
            return deco
        self._map[obj.__name__] = obj
        return obj

    def get(self, name):
        return self._map[name]


def _configurable(init_func=None, *, from_config=None):
    # explicit-kwargs construction only (we never pass a cfg)
    if init_func is not None:
        return init_func

    def deco(f):
        return f
    return deco


class _ShapeSpec:
    def __init__(self, channels=None, height=None, width=None, stride=None):
        self.channels, self.height, self.width, self.stride = channels, height, width, stride


class _Conv2d(nn.Conv2d):
    def __init__(self, *args, **kwargs):
        norm = kwargs.pop("norm", None)
        activation = kwargs.pop("activation", None)
        super().__init__(*args, **kwargs)
        self.norm = norm
        self.activation = activation

    def forward(self, x):
        x = F.conv2d(x, self.weight, self.bias, self.stride, self.padding, self.dilation, self.groups)
        if self.norm is not None:
            x = self.norm(x)
        if self.activation is not None:
            x = self.activation(x)
        return x


def _get_norm(norm, out_channels):
    if norm is None or norm == "":
        return None
    if norm == "GN":
        return nn.GroupNorm(32, out_channels)
    if norm == "LN":
        return nn.GroupNorm(1, out_channels)
    raise ValueError(norm)


def _c2_xavier_fill(module):
    nn.init.kaiming_uniform_(module.weight, a=1)
    if module.bias is not None:
        nn.init.constant_(module.bias, 0)


def _c2_msra_fill(module):
    nn.init.kaiming_normal_(module.weight, mode="fan_out", nonlinearity="relu")
    if module.bias is not None:
        nn.init.constant_(module.bias, 0)


def point_sample(input, point_coords, **kwargs):
    add_dim = False
    if point_coords.dim() == 3:
        add_dim = True
        point_coords = point_coords.unsqueeze(2)
    output = F.grid_sample(input, 2.0 * point_coords - 1.0, **kwargs)
    if add_dim:
        output = output.squeeze(3)
    return output


TOPK_RECORD = None  # gen_golden.py sets this to a list: every call appends its chosen top-k indices [num_boxes, k]


def get_uncertain_point_coords_with_randomness(coarse_logits, uncertainty_func, num_points, oversample_ratio, importance_sample_ratio):
    assert oversample_ratio >= 1
    assert 0 <= importance_sample_ratio <= 1
    num_boxes = coarse_logits.shape[0]
    num_sampled = int(num_points * oversample_ratio)
    point_coords = torch.rand(num_boxes, num_sampled, 2, device=coarse_logits.device)
    point_logits = point_sample(coarse_logits, point_coords, align_corners=False)
    point_uncertainties = uncertainty_func(point_logits)
    num_uncertain_points = int(importance_sample_ratio * num_points)
    num_random_points = num_points - num_uncertain_points
    idx = torch.topk(point_uncertainties[:, 0, :], k=num_uncertain_points, dim=1)[1]
    if TOPK_RECORD is not None:
        TOPK_RECORD.append(idx.clone())
    shift = num_sampled * torch.arange(num_boxes, dtype=torch.long, device=coarse_logits.device)
    idx = idx + shift[:, None]
    point_coords = point_coords.view(-1, 2)[idx.view(-1), :].view(num_boxes, num_uncertain_points, 2)
    if num_random_points > 0:
        point_coords = torch.cat(
            [point_coords, torch.rand(num_boxes, num_random_points, 2, device=coarse_logits.device)], dim=1
        )
    return point_coords


class _DropPath(nn.Module):
    def __init__(self, p=0.0):
        super().__init__()
        assert p == 0.0

    def forward(self, x):
        return x


_installed = False


def install():
    """Install the stand-ins and namespace packages.  Idempotent."""
    global _installed
    if _installed:
        return
    if not os.path.isdir(REF_ROOT):
        raise RuntimeError("reference checkout not present at /root/reference: golden vectors can only be "
                           "(re)generated in the build container")
    # --- third-party stand-ins -------------------------------------------------------------
    _mod("detectron2")
    _mod("detectron2.config", configurable=_configurable)
    _mod("detectron2.layers", Conv2d=_Conv2d, ShapeSpec=_ShapeSpec, get_norm=_get_norm, DeformConv=object)
    _mod("detectron2.modeling", SEM_SEG_HEADS_REGISTRY=_Registry("SEM_SEG_HEADS"))
    _mod("detectron2.utils")
    _mod("detectron2.utils.registry", Registry=_Registry)
    _mod("detectron2.utils.comm", get_world_size=lambda: 1)
    _mod("detectron2.projects")
    _mod("detectron2.projects.point_rend")
    _mod("detectron2.projects.point_rend.point_features", point_sample=point_sample,
         get_uncertain_point_coords_with_randomness=get_uncertain_point_coords_with_randomness)
    _mod("fvcore")
    _mod("fvcore.nn")
    _mod("fvcore.nn.weight_init", c2_xavier_fill=_c2_xavier_fill, c2_msra_fill=_c2_msra_fill)
    _mod("timm")
    _mod("timm.models")
    _mod("timm.models.layers", DropPath=_DropPath)
    if "torchvision" not in sys.modules:
        _mod("torchvision", _is_tracing=lambda: False)
    # empty native module: ms_deform_attn_func.py:21 imports it, ms_deform_attn.py:123 then falls back
    _mod("MultiScaleDeformableAttention")
    # --- reference packages as bare namespaces (never run models/__init__.py) ---------------
    for name, rel in (("models", "models"), ("models.modeling", "models/modeling"), ("models.utils", "models/utils")):
        m = _mod(name)
        m.__path__ = [os.path.join(REF_ROOT, rel)]
    # criterion.py calls .cuda() unconditionally (criterion.py:218,225,244): identity on this CPU box
    torch.Tensor.cuda = lambda self, *a, **k: self
    _installed = True


def ref():
    """Return a namespace with the reference's hot-path classes/functions."""
    install()
    import importlib
    ns = types.SimpleNamespace()
    pe = importlib.import_module("models.modeling.transformer_decoder.position_encoding")
    fh = importlib.import_module("models.modeling.fusion_module.utils.fuse_helper")
    av = importlib.import_module("models.modeling.fusion_module.AVFuse")
    mf = importlib.import_module("models.modeling.pixel_decoder.ops.functions.ms_deform_attn_func")
    mm = importlib.import_module("models.modeling.pixel_decoder.ops.modules.ms_deform_attn")
    pd = importlib.import_module("models.modeling.pixel_decoder.msdeformattn")
    td = importlib.import_module("models.modeling.transformer_decoder.transformer_decoder")
    at = importlib.import_module("models.modeling.misc.audio_transformation")
    hd = importlib.import_module("models.modeling.meta_arch.mask_former_head")
    ms = importlib.import_module("models.utils.misc")
    mt = importlib.import_module("models.modeling.matcher")
    cr = importlib.import_module("models.modeling.criterion")
    cs = importlib.import_module("models.modeling.criterion_ss")
    ns.PositionEmbeddingSine = pe.PositionEmbeddingSine
    ns.BiAttentionBlock = fh.BiAttentionBlock
    ns.BiMultiHeadAttention = fh.BiMultiHeadAttention
    ns.AVFuse = av.AVFuse
    ns.ms_deform_attn_core_pytorch = mf.ms_deform_attn_core_pytorch
    ns.MSDeformAttn = mm.MSDeformAttn
    ns.MSDeformAttnPixelDecoder = pd.MSDeformAttnPixelDecoder
    ns.MultiScaleMaskedTransformerDecoder = td.MultiScaleMaskedTransformerDecoder
    ns.audio_mlp = at.audio_mlp
    ns.MaskFormerHead = hd.MaskFormerHead
    ns.channel_weighted_block = ms.channel_weighted_block
    ns.HungarianMatcher = mt.HungarianMatcher
    ns.SetCriterion = cr.SetCriterion
    ns.SetCriterion_SS = cs.SetCriterion_SS
    ns.ShapeSpec = _ShapeSpec
    return ns


def build_head(num_classes=2, channels=(256, 512, 1024, 2048), num_queries=100, dec_layers=9, enc_layers=6,
               use_cosine_loss=True, num_frames=5, dataset_name="avss4"):
    """Construct the reference MaskFormerHead exactly as COMBO_R50_bs8_90k.yaml would
    (mask_former_head.py:94-136, msdeformattn.py:299-313, transformer_decoder.py:366-403)."""
    R = ref()
    shapes = {f"res{i + 2}": _ShapeSpec(channels=c, stride=s) for i, (c, s) in enumerate(zip(channels, (4, 8, 16, 32)))}
    pixel_decoder = R.MSDeformAttnPixelDecoder(
        shapes, transformer_dropout=0.0, transformer_nheads=8, transformer_dim_feedforward=1024,
        transformer_enc_layers=enc_layers, conv_dim=256, mask_dim=256, norm="GN",
        transformer_in_features=["res3", "res4", "res5"], common_stride=4)
    fusion = R.AVFuse(fused_type="MHA-B", audio_dim=128, fused_backbone=["res2"], fused_backbone_dim=[256])
    amlp = R.audio_mlp(in_dim=128, middle_dim=4096, out_dim=256)
    predictor = R.MultiScaleMaskedTransformerDecoder(
        256, True, num_classes=num_classes, hidden_dim=256, num_queries=num_queries, num_frames=num_frames,
        queries_fuse_type="add", audio_out_dim=256, nheads=8, dim_feedforward=2048, dec_layers=dec_layers,
        pre_norm=False, mask_dim=256, enforce_input_project=False, dataset_name=dataset_name,
        use_cosine_loss=use_cosine_loss)
    head = R.MaskFormerHead(shapes, num_classes=num_classes, pixel_decoder=pixel_decoder, fusion_module=fusion,
                            audio_transformation=amlp, loss_weight=1.0, ignore_value=255,
                            transformer_predictor=predictor, transformer_in_feature="multi_scale_pixel_decoder")
    return head
