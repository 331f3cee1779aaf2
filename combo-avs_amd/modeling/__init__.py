from .pixel_decoder import MSDeformAttnPixelDecoder, MSDeformAttn  # noqa: F401
from .fusion import AVFuse  # noqa: F401
from .transformer_decoder import MultiScaleMaskedTransformerDecoder  # noqa: F401
from .head import MaskFormerHead, audio_mlp  # noqa: F401
from .criterion import SetCriterion, SetCriterion_SS  # noqa: F401
from .matcher import HungarianMatcher  # noqa: F401
