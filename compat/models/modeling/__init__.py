"""models.modeling of the reference: importing it registers the head / pixel decoder / transformer decoder (here: combo_avs_amd's)."""
from combo_avs_amd.modeling import criterion, fusion, head, matcher, pixel_decoder, transformer_decoder  # noqa: F401
