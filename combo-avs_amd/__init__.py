"""combo-avs_amd: MI355X-native (gfx950) implementation of COMBO-AVS's audio-visual fusion and
mask-decoding hot path.  HIP kernels + C ABI live in `csrc/` (-> lib/libcombo_avs_hip.so, declared in
include/combo_avs.h); the Python here mirrors the reference's module/operator interface for that path
and is plumbing only (device memory, streams, autograd bookkeeping, torch.distributed).

There is NO CPU fallback: every op that has a HIP kernel calls it through the C ABI and raises if the
library is missing (see _lib.py).
"""
__version__ = "0.1.0"

from . import _lib  # noqa: F401

from .config import (CfgNode, add_audio_config, add_fuse_config, add_maskformer2_config, combo_cfg, get_cfg)  # noqa: E402,F401
from .registry import (BACKBONE_REGISTRY, META_ARCH_REGISTRY, SEM_SEG_HEADS_REGISTRY,  # noqa: E402,F401
                       TRANSFORMER_DECODER_REGISTRY)
from . import backbone, backbone_pvt  # noqa: E402,F401  (populate BACKBONE_REGISTRY: build_resnet_backbone, build_pvtv2_b5_backbone)

from . import d2_register  # noqa: E402,F401  (explicit: d2_register.install(); importing this package never touches detectron2's registries)
