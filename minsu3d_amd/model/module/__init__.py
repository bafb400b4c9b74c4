from .common import ResidualBlock, UBlock  # noqa: F401
from .networks import Backbone, TinyUnet, PointLinear  # noqa: F401
