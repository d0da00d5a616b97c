"""deflow_amd -- MI355X-native DeFlow scene-flow hot path (HIP kernels behind the reference's model plugin API)."""
from .deflow import DeFlow, cal_pose0to1  # noqa: F401
from .encoder import DynamicEmbedder  # noqa: F401
from .unet import ConvWithNorms, FastFlow3DUNet  # noqa: F401
from .decoder import ConvGRU, ConvGRUDecoder, LinearDecoder  # noqa: F401
from .timer import Timing  # noqa: F401
