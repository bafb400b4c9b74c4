from .common import ResidualBlock, UBlock  # noqa: F401
from .backbone import Backbone  # noqa: F401
from .tiny_unet import TinyUnet  # noqa: F401
