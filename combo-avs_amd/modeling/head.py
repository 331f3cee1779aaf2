"""MaskFormerHead (mirrors meta_arch/mask_former_head.py:18-159) and audio_mlp (misc/audio_transformation.py:5-14)."""
import logging
from typing import Dict

from torch import nn

from ..registry import SEM_SEG_HEADS_REGISTRY, TRANSFORMER_DECODER_REGISTRY, ShapeSpec
from .fusion import AVFuse


class audio_mlp(nn.Module):
    def __init__(self, in_dim=128, middle_dim=4096, out_dim=256):
        super().__init__()
        self.embeddings = nn.Sequential(nn.Linear(in_dim, middle_dim), nn.ReLU(True), nn.Linear(middle_dim, middle_dim),
                                        nn.ReLU(True), nn.Linear(middle_dim, out_dim))

    def forward(self, x):
        return self.embeddings(x)


def build_pixel_decoder(cfg, input_shape):
    name = cfg.MODEL.SEM_SEG_HEAD.PIXEL_DECODER_NAME
    cls = SEM_SEG_HEADS_REGISTRY.get(name)
    model = cls(**cls.from_config(cfg, input_shape))
    if not callable(getattr(model, "forward_features", None)):
        raise ValueError("Only SEM_SEG_HEADS with forward_features method can be used as pixel decoder. "
                         f"Please implement forward_features for {name} to only return mask features.")
    return model


def build_transformer_decoder(cfg, in_channels, mask_classification=True):
    cls = TRANSFORMER_DECODER_REGISTRY.get(cfg.MODEL.MASK_FORMER.TRANSFORMER_DECODER_NAME)
    return cls(**cls.from_config(cfg, in_channels, mask_classification))


@SEM_SEG_HEADS_REGISTRY.register()
class MaskFormerHead(nn.Module):
    _version = 2

    def _load_from_state_dict(self, state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys, error_msgs):
        version = local_metadata.get("version", None)
        if version is None or version < 2:  # mask_former_head.py:22-42
            for k in list(state_dict.keys()):
                if "sem_seg_head" in k and not k.startswith(prefix + "predictor"):
                    newk = k.replace(prefix, prefix + "pixel_decoder.")
                    if newk != k:
                        state_dict[newk] = state_dict.pop(k)
                        logging.getLogger(__name__).warning("Weight format of %s have changed!", self.__class__.__name__)
        super()._load_from_state_dict(state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys, error_msgs)

    def __init__(self, input_shape: Dict[str, ShapeSpec], *, num_classes: int, pixel_decoder: nn.Module,
                 fusion_module, audio_transformation, loss_weight: float = 1.0, ignore_value: int = -1,
                 transformer_predictor: nn.Module, transformer_in_feature: str):
        super().__init__()
        input_shape = sorted(input_shape.items(), key=lambda x: x[1].stride)
        self.in_features = [k for k, v in input_shape]
        self.ignore_value = ignore_value
        self.common_stride = 4
        self.loss_weight = loss_weight
        self.pixel_decoder = pixel_decoder
        if fusion_module is not None:
            self.late_fusion = True
            self.fusion_module = fusion_module
            self.audio_transformation = audio_transformation
        else:
            self.late_fusion = False
        self.predictor = transformer_predictor
        self.transformer_in_feature = transformer_in_feature
        self.num_classes = num_classes

    @classmethod
    def from_config(cls, cfg, input_shape: Dict[str, ShapeSpec]):
        tif = cfg.MODEL.MASK_FORMER.TRANSFORMER_IN_FEATURE
        if tif in ("transformer_encoder", "multi_scale_pixel_decoder"):
            in_ch = cfg.MODEL.SEM_SEG_HEAD.CONVS_DIM
        elif tif == "pixel_embedding":
            in_ch = cfg.MODEL.SEM_SEG_HEAD.MASK_DIM
        else:
            in_ch = input_shape[tif].channels
        if cfg.MODEL.FUSE_CONFIG.FUSION_STEP == "late":
            audio_out_dim = 128 if cfg.MODEL.FUSE_CONFIG.QUERIES_FUSE_TYPE == "dim" else 256
            cfg.defrost()
            cfg.MODEL.FUSE_CONFIG.AUDIO_OUT_DIM = audio_out_dim  # the reference mutates the cfg too (:112-114)
            cfg.freeze()
            fusion_module = AVFuse(**AVFuse.from_config(cfg))
            audio_transformation = audio_mlp(in_dim=128, middle_dim=4096, out_dim=audio_out_dim)
        else:
            fusion_module = audio_transformation = None
        return dict(
            input_shape={k: v for k, v in input_shape.items() if k in cfg.MODEL.SEM_SEG_HEAD.IN_FEATURES},
            ignore_value=cfg.MODEL.SEM_SEG_HEAD.IGNORE_VALUE, num_classes=cfg.MODEL.SEM_SEG_HEAD.NUM_CLASSES,
            pixel_decoder=build_pixel_decoder(cfg, input_shape), fusion_module=fusion_module,
            audio_transformation=audio_transformation, loss_weight=cfg.MODEL.SEM_SEG_HEAD.LOSS_WEIGHT,
            transformer_in_feature=tif,
            transformer_predictor=build_transformer_decoder(cfg, in_ch, mask_classification=True))

    def forward(self, features, audio_features, mask=None):
        return self.layers(features, audio_features, mask)

    def layers(self, features, audio_feature, mask=None):
        mask_features, _, multi_scale_features = self.pixel_decoder.forward_features(features)
        if self.late_fusion:
            fused = self.fusion_module({"res2": mask_features}, audio_feature)  # mask_former_head.py:144-151
            mask_features = fused["visual"]["res2"]
            audio_feature = self.audio_transformation(fused["audio"])
        if self.transformer_in_feature != "multi_scale_pixel_decoder":
            raise NotImplementedError("only TRANSFORMER_IN_FEATURE == 'multi_scale_pixel_decoder' (all shipped configs)")
        return self.predictor(multi_scale_features, audio_feature, mask_features, mask)
