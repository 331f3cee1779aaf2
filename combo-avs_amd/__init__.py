"""combo-avs_amd: MI355X-native (gfx950) implementation of COMBO-AVS's audio-visual fusion and
mask-decoding hot path.  HIP kernels + C ABI live in `csrc/` (-> lib/libcombo_avs_hip.so, declared in
include/combo_avs.h); the Python here mirrors the reference's module/operator interface for that path
and is plumbing only (device memory, streams, autograd bookkeeping, torch.distributed).

There is NO CPU fallback: every op that has a HIP kernel calls it through the C ABI and raises if the
library is missing (see _lib.py).
"""
__version__ = "0.1.0"

import os as _os


def _guard_graph_memsets():
    """hipGraph memset nodes replay wrongly with the HIP runtime's AQL packet capture on this stack (ROCm 7.2 / torch 2.10:
    tools/graph_reduce_repro.py - stale results in 99 of 100 replays, none with DEBUG_CLR_GRAPH_PACKET_CAPTURE=0; measured cost
    of switching it off on the captured training step: none, 55.98 vs 56.46 ms).  This package's own ops put no memset node
    into the captured step (ops/colsum.py), but library calls may (MIOpen's backward of the stride-2 3x3 convolutions does:
    tools/memset_sites.py), so the switch is turned off here unless the caller has set it.  The runtime reads it at its first
    HIP call, which may already lie behind us - trainer.GraphedTrainStep therefore TESTS the behaviour before it captures
    (trainer.graph_memset_selftest) instead of trusting this."""
    if "DEBUG_CLR_GRAPH_PACKET_CAPTURE" in _os.environ:
        return "user"
    if _os.environ.get("COMBO_GRAPH_MEMSET_GUARD", "1") == "0":  # the embedding application manages the runtime's switches itself
        return "off"
    _os.environ["DEBUG_CLR_GRAPH_PACKET_CAPTURE"] = "0"
    import logging
    logging.getLogger(__name__).info(
        "combo_avs_amd: set DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 for this process and its children (hipGraph memset nodes, "
        "INTEGRATION.md section 6); COMBO_GRAPH_MEMSET_GUARD=0 leaves the environment alone")
    return "set"


GRAPH_MEMSET_GUARD = _guard_graph_memsets()

from . import _lib  # noqa: E402,F401

from .config import (CfgNode, add_audio_config, add_fuse_config, add_maskformer2_config, combo_cfg, get_cfg)  # noqa: E402,F401
from .registry import (BACKBONE_REGISTRY, META_ARCH_REGISTRY, SEM_SEG_HEADS_REGISTRY,  # noqa: E402,F401
                       TRANSFORMER_DECODER_REGISTRY)
from . import backbone, backbone_pvt  # noqa: E402,F401  (populate BACKBONE_REGISTRY: build_resnet_backbone, build_pvtv2_b5_backbone)

from . import d2_register  # noqa: E402,F401  (explicit: d2_register.install(); importing this package never touches detectron2's registries)
