"""CPU: drop-in boundary: registration into detectron2-shaped registries under the reference's names and
the `models` alias package the reference's drivers import from (train_net.py:52-62)."""
import importlib
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class D2Registry:
    """fvcore.common.registry.Registry's surface (detectron2 is not installed here): _obj_map, register(obj), get, __contains__"""

    def __init__(self, name):
        self._name, self._obj_map = name, {}

    def register(self, obj=None):
        assert obj is not None
        name = obj.__name__
        assert name not in self._obj_map, f"An object named '{name}' was already registered in '{self._name}' registry!"
        self._obj_map[name] = obj
        return obj

    def get(self, name):
        return self._obj_map[name]

    def __contains__(self, name):
        return name in self._obj_map


def _registries():
    return {n: D2Registry(n) for n in ("META_ARCH_REGISTRY", "SEM_SEG_HEADS_REGISTRY", "BACKBONE_REGISTRY")}


def test_install_registers_the_reference_names_and_respects_existing_entries():
    import combo_avs_amd  # noqa: F401
    from combo_avs_amd import d2_register
    regs = _registries()

    def MaskFormer(cfg):  # what the reference's own `models` package would have registered
        return "reference"
    regs["META_ARCH_REGISTRY"].register(MaskFormer)
    done = d2_register.install(regs, override=False)
    assert "META_ARCH_REGISTRY[MaskFormer]" not in done and regs["META_ARCH_REGISTRY"].get("MaskFormer") is MaskFormer
    assert {"SEM_SEG_HEADS_REGISTRY[MaskFormerHead]", "SEM_SEG_HEADS_REGISTRY[MSDeformAttnPixelDecoder]",
            "BACKBONE_REGISTRY[build_pvtv2_b5_backbone]"} <= set(done)
    done = d2_register.install(regs, override=True)
    assert "META_ARCH_REGISTRY[MaskFormer]" in done
    from combo_avs_amd.meta_arch import MaskFormer as Ours
    assert regs["META_ARCH_REGISTRY"].get("MaskFormer").combo_class is Ours


def test_registered_factories_build_from_a_cfg_like_detectron2_build_model_does():
    """detectron2: `META_ARCH_REGISTRY.get(cfg.MODEL.META_ARCHITECTURE)(cfg)`, `SEM_SEG_HEADS_REGISTRY.get(name)(cfg, shape)`"""
    import combo_avs_amd  # noqa: F401
    from combo_avs_amd import combo_cfg, d2_register
    from combo_avs_amd.backbone import ResNet
    regs = _registries()
    d2_register.install(regs, override=True)
    cfg = combo_cfg(os.path.join(ROOT, "configs/avs_s4/COMBO_R50_bs8_90k.yaml"))
    head = regs["SEM_SEG_HEADS_REGISTRY"].get(cfg.MODEL.SEM_SEG_HEAD.NAME)(cfg, ResNet(50).output_shape())
    assert type(head).__name__ == "MaskFormerHead" and hasattr(head, "pixel_decoder") and hasattr(head, "predictor")
    model = regs["META_ARCH_REGISTRY"].get(cfg.MODEL.META_ARCHITECTURE)(cfg)
    assert type(model).__name__ == "MaskFormer"
    keys = set(model.state_dict().keys())
    assert "sem_seg_head.predictor.query_feat.weight" in keys and "scale_factor_module.0.fc1.weight" in keys


def test_models_alias_package_serves_the_drivers_import_list():
    """`from models import (...)` of train_net.py:52-62 in a child process with <repo>/compat first on the path."""
    code = (
        "import models\n"
        "from models import add_maskformer2_config, add_audio_config, add_fuse_config, MaskFormer\n"
        "from models.config import add_fuse_config as f2\n"
        "from models.maskformer_model import MaskFormer as M2\n"
        "import models.modeling\n"
        "import combo_avs_amd\n"
        "assert M2 is MaskFormer and MaskFormer is combo_avs_amd.meta_arch.MaskFormer and f2 is add_fuse_config\n"
        "cfg = combo_avs_amd.get_cfg(); add_audio_config(cfg); add_fuse_config(cfg); add_maskformer2_config(cfg)\n"
        "assert cfg.MODEL.MASK_FORMER.NUM_OBJECT_QUERIES == 100 and cfg.MODEL.FUSE_CONFIG.TYPE == 'MHA-B'\n"
        "try:\n"
        "    models.BestCheckpointer\n"
        "    raise SystemExit('expected ImportError without COMBO_REFERENCE_ROOT / detectron2')\n"
        "except ImportError as e:\n"
        "    assert 'COMBO_REFERENCE_ROOT' in str(e)\n"
        "print('ok')\n")
    env = dict(os.environ, PYTHONPATH=os.path.join(ROOT, "compat") + os.pathsep + ROOT)
    env.pop("COMBO_REFERENCE_ROOT", None)
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "ok" in r.stdout, r.stderr[-1500:]


def test_add_config_functions_extend_a_foreign_cfg_node_type():
    """the add_* functions create sub-nodes of the cfg's OWN node type (detectron2's CfgNode in the drivers)"""
    import combo_avs_amd  # noqa: F401
    from combo_avs_amd.config import CfgNode, add_audio_config, add_fuse_config, add_maskformer2_config, get_cfg

    class ForeignNode(CfgNode):
        pass
    base = get_cfg()
    cfg = ForeignNode(base)
    cfg.MODEL = ForeignNode(base.MODEL)
    cfg.MODEL.SEM_SEG_HEAD = ForeignNode(base.MODEL.SEM_SEG_HEAD)
    cfg.INPUT = ForeignNode(base.INPUT)
    cfg.INPUT.CROP = ForeignNode(base.INPUT.CROP)
    cfg.SOLVER = ForeignNode(base.SOLVER)
    add_audio_config(cfg), add_fuse_config(cfg), add_maskformer2_config(cfg)
    assert type(cfg.MODEL.MASK_FORMER) is ForeignNode and type(cfg.MODEL.FUSE_CONFIG) is ForeignNode and type(cfg.MODEL.AUDIO) is ForeignNode
