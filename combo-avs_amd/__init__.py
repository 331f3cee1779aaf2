"""combo-avs_amd: MI355X-native (gfx950) implementation of COMBO-AVS's audio-visual fusion and
mask-decoding hot path.  HIP kernels + C ABI live in `csrc/` (-> lib/libcombo_avs_hip.so, declared in
include/combo_avs.h); the Python here mirrors the reference's module/operator interface for that path
and is plumbing only (device memory, streams, autograd bookkeeping, torch.distributed).

There is NO CPU fallback: every op that has a HIP kernel calls it through the C ABI and raises if the
library is missing (see _lib.py).
"""
__version__ = "0.1.0"

from . import _lib  # noqa: F401
