"""models.maskformer_model of the reference (maskformer_model.py:28): the meta-architecture, from combo_avs_amd."""
from combo_avs_amd.meta_arch import MaskFormer  # noqa: F401
