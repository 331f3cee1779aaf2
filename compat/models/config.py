"""models.config of the reference (config.py:6-149): the three add_* functions, from combo_avs_amd.config."""
from combo_avs_amd.config import add_audio_config, add_fuse_config, add_maskformer2_config  # noqa: F401
