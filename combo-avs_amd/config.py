"""Config surface of the reference (models/config.py + detectron2 defaults it relies on), as a small yacs-like
CfgNode.  Key names are the reference's (SURVEY §8(b) "config surface"); YAML files use `_BASE_` inheritance and
the reference's one `!!python/object/apply:eval` tag (R50-AVSS4-SemanticSegmentation.yaml:48) is special-cased
to the list it evaluates to instead of being executed."""
import copy
import os

import yaml


class CfgNode(dict):
    def __init__(self, init=None):
        super().__init__()
        object.__setattr__(self, "_frozen", False)
        for k, v in (init or {}).items():
            self[k] = CfgNode(v) if isinstance(v, dict) and not isinstance(v, CfgNode) else v

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    def __setattr__(self, k, v):
        if self._frozen:
            raise AttributeError(f"Attempted to set {k} to {v}, but CfgNode is immutable")
        self[k] = v

    def freeze(self):
        self._set_frozen(True)

    def defrost(self):
        self._set_frozen(False)

    def _set_frozen(self, f):
        object.__setattr__(self, "_frozen", f)
        for v in self.values():
            if isinstance(v, CfgNode):
                v._set_frozen(f)

    def clone(self):
        return copy.deepcopy(self)

    def merge_from_other_cfg(self, other):
        for k, v in other.items():
            if isinstance(v, dict) and isinstance(self.get(k), CfgNode):
                self[k].merge_from_other_cfg(v)
            else:
                self[k] = CfgNode(v) if isinstance(v, dict) else (list(v) if isinstance(v, tuple) else v)

    def merge_from_file(self, path):
        self.merge_from_other_cfg(load_yaml_with_base(path))

    def merge_from_list(self, opts):
        assert len(opts) % 2 == 0
        for k, v in zip(opts[0::2], opts[1::2]):
            node = self
            parts = k.split(".")
            for p in parts[:-1]:
                node = node[p]
            if isinstance(v, str):
                try:
                    v = yaml.safe_load(v)
                except yaml.YAMLError:
                    pass
            node[parts[-1]] = v


class _Loader(yaml.SafeLoader):
    pass


def _apply_eval(loader, node):
    # the only use in the reference: eval ["[int(x * 0.1 * 224) for x in range(5, 21)]"]
    args = loader.construct_sequence(node)
    if args == ["[int(x * 0.1 * 224) for x in range(5, 21)]"]:
        return [int(x * 0.1 * 224) for x in range(5, 21)]
    raise ValueError(f"refusing to eval {args!r} from a config file")


_Loader.add_constructor("tag:yaml.org,2002:python/object/apply:eval", _apply_eval)


def load_yaml_with_base(path):
    with open(path) as f:
        cfg = yaml.load(f, Loader=_Loader) or {}
    base = cfg.pop("_BASE_", None)
    if base is not None:
        if not os.path.isabs(base):
            base = os.path.join(os.path.dirname(path), base)
        merged = CfgNode(load_yaml_with_base(base))
        merged.merge_from_other_cfg(cfg)
        return merged
    return cfg


def get_cfg():
    """The subset of detectron2's defaults the hot path reads (+ add_deeplab_config's solver keys)."""
    C = CfgNode
    cfg = C({
        "VERSION": 2, "OUTPUT_DIR": "./output", "SEED": -1,
        "MODEL": {
            "META_ARCHITECTURE": "MaskFormer", "DEVICE": "cuda", "WEIGHTS": "",
            "PIXEL_MEAN": [123.675, 116.280, 103.530], "PIXEL_STD": [58.395, 57.120, 57.375],
            "BACKBONE": {"NAME": "build_resnet_backbone", "FREEZE_AT": 0},
            "RESNETS": {"DEPTH": 50, "OUT_FEATURES": ["res2", "res3", "res4", "res5"], "NORM": "FrozenBN",
                        "STEM_TYPE": "basic", "STEM_OUT_CHANNELS": 64, "STRIDE_IN_1X1": False, "RES5_MULTI_GRID": [1, 1, 1]},
            "SEM_SEG_HEAD": {"NAME": "MaskFormerHead", "IN_FEATURES": ["res2", "res3", "res4", "res5"], "IGNORE_VALUE": 255,
                             "NUM_CLASSES": 2, "CONVS_DIM": 256, "COMMON_STRIDE": 4, "NORM": "GN", "LOSS_WEIGHT": 1.0},
        },
        "DATASETS": {"TRAIN": ("avss4_sem_seg_train",), "TEST": ("avss4_sem_seg_val",)},
        "DATALOADER": {"NUM_WORKERS": 8, "FILTER_EMPTY_ANNOTATIONS": True},
        "INPUT": {"MIN_SIZE_TRAIN": [224], "MIN_SIZE_TRAIN_SAMPLING": "choice", "MAX_SIZE_TRAIN": 896, "MIN_SIZE_TEST": 224,
                  "MAX_SIZE_TEST": 896, "FORMAT": "RGB", "CROP": {"ENABLED": True, "TYPE": "absolute", "SIZE": [224, 224]}},
        "SOLVER": {"IMS_PER_BATCH": 8, "BASE_LR": 0.0001, "MAX_ITER": 90000, "WARMUP_FACTOR": 1.0, "WARMUP_ITERS": 0,
                   "WEIGHT_DECAY": 0.05, "WEIGHT_DECAY_NORM": 0.0, "LR_SCHEDULER_NAME": "WarmupPolyLR", "MOMENTUM": 0.9,
                   "POLY_LR_POWER": 0.9, "POLY_LR_CONSTANT_ENDING": 0.0,
                   "CLIP_GRADIENTS": {"ENABLED": True, "CLIP_TYPE": "full_model", "CLIP_VALUE": 0.01, "NORM_TYPE": 2.0},
                   "AMP": {"ENABLED": False}},
        "TEST": {"EVAL_PERIOD": 5000, "DETECTIONS_PER_IMAGE": 100, "AUG": {"ENABLED": False}},
    })
    return cfg


def add_audio_config(cfg):  # models/config.py:6-12
    CN = type(cfg)  # sub-nodes of the cfg's own node type: the same function extends a detectron2 CfgNode (d2_register.py)
    cfg.MODEL.AUDIO = CN({
        "FREEZE_AUDIO_EXTRACTOR": True, "PRETRAINED_VGGISH_MODEL_PATH": "./torchvggish/vggish-10086976.pth",
        "PREPROCESS_AUDIO_TO_LOG_MEL": True, "POSTPROCESS_LOG_MEL_WITH_PCA": False,
        "PRETRAINED_PCA_PARAMS_PATH": "./torchvggish/vggish_pca_params-970ea276.pth"})


def add_fuse_config(cfg):  # models/config.py:15-32
    CN = type(cfg)
    cfg.MODEL.FUSE_CONFIG = CN({
        "FUSION_STEP": "early", "TYPE": "MHA-B", "AUDIO_DIM": 1024, "FUSED_BACKBONE": [], "FUSED_BACKBONE_DIM": [],
        "NUM_FRAMES": 5, "QUERIES_FUSE_TYPE": "add", "AUDIO_OUT_DIM": 256})
    cfg.MODEL.MOBILE_SAM = CN({"USE_MOBILE_SAM": False, "CHECKPOINT": ""})
    cfg.MODEL.PRE_SAM = CN({"USE_PRE_SAM": False, "PRE_SAM_DIM": [256, 512, 1024, 2048],
                                 "PRE_SAM_FEATURE_SIZE": [56, 28, 14, 7]})


def add_maskformer2_config(cfg):  # models/config.py:35-149
    CN = type(cfg)
    cfg.INPUT.AUGMENTATION = True
    cfg.INPUT.DATASET_MAPPER_NAME = "mask_former_semantic"
    cfg.INPUT.COLOR_AUG_SSD = False
    cfg.INPUT.CROP.SINGLE_CATEGORY_MAX_AREA = 1.0
    cfg.INPUT.SIZE_DIVISIBILITY = -1
    cfg.SOLVER.WEIGHT_DECAY_EMBED = 0.0
    cfg.SOLVER.OPTIMIZER = "ADAMW"
    cfg.SOLVER.BACKBONE_MULTIPLIER = 0.1
    cfg.MODEL.MASK_FORMER = CN({
        "DEEP_SUPERVISION": True, "NO_OBJECT_WEIGHT": 0.1, "CLASS_WEIGHT": 1.0, "DICE_WEIGHT": 1.0, "MASK_WEIGHT": 20.0,
        "COSINE_WEIGHT": 1.0, "NHEADS": 8, "DROPOUT": 0.1, "DIM_FEEDFORWARD": 2048, "ENC_LAYERS": 0, "DEC_LAYERS": 6,
        "PRE_NORM": False, "HIDDEN_DIM": 256, "NUM_OBJECT_QUERIES": 100, "TRANSFORMER_IN_FEATURE": "res5",
        "ENFORCE_INPUT_PROJ": False,
        "TEST": {"SEMANTIC_ON": True, "INSTANCE_ON": False, "PANOPTIC_ON": False, "OBJECT_MASK_THRESHOLD": 0.0,
                 "OVERLAP_THRESHOLD": 0.0, "SEM_SEG_POSTPROCESSING_BEFORE_INFERENCE": False},
        "SIZE_DIVISIBILITY": 32, "TRANSFORMER_DECODER_NAME": "MultiScaleMaskedTransformerDecoder",
        "TRAIN_NUM_POINTS": 112 * 112, "OVERSAMPLE_RATIO": 3.0, "IMPORTANCE_SAMPLE_RATIO": 0.75})
    cfg.MODEL.SEM_SEG_HEAD.MASK_DIM = 256
    cfg.MODEL.SEM_SEG_HEAD.TRANSFORMER_ENC_LAYERS = 0
    cfg.MODEL.SEM_SEG_HEAD.PIXEL_DECODER_NAME = "BasePixelDecoder"
    cfg.MODEL.PVT = CN({"OUT_FEATURES": ["res2", "res3", "res4", "res5"]})
    cfg.INPUT.IMAGE_SIZE = 1024
    cfg.INPUT.MIN_SCALE = 0.1
    cfg.INPUT.MAX_SCALE = 2.0
    cfg.MODEL.SEM_SEG_HEAD.DEFORMABLE_TRANSFORMER_ENCODER_IN_FEATURES = ["res3", "res4", "res5"]
    cfg.MODEL.SEM_SEG_HEAD.DEFORMABLE_TRANSFORMER_ENCODER_N_POINTS = 4
    cfg.MODEL.SEM_SEG_HEAD.DEFORMABLE_TRANSFORMER_ENCODER_N_HEADS = 8


def combo_cfg(config_file=None, opts=()):
    """get_cfg + the three add_* calls + file + opts, as train_net.py:231-247 of the reference does."""
    cfg = get_cfg()
    add_audio_config(cfg)
    add_fuse_config(cfg)
    add_maskformer2_config(cfg)
    if config_file:
        cfg.merge_from_file(config_file)
    cfg.merge_from_list(list(opts))
    cfg.freeze()
    return cfg
