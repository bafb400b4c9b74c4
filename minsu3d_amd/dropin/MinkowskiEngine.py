"""`import MinkowskiEngine as ME` for the reference's call sites (SparseTensor,
MinkowskiConvolution, MinkowskiConvolutionTranspose, MinkowskiBatchNorm, MinkowskiReLU, cat,
utils.sparse_quantize, utils.sparse_collate) -- an alias of minsu3d_amd.MinkowskiEngine."""
import sys

import minsu3d_amd.MinkowskiEngine as _me
from minsu3d_amd.MinkowskiEngine import *  # noqa: F401,F403
from minsu3d_amd.MinkowskiEngine import utils  # noqa: F401

sys.modules.setdefault("MinkowskiEngine.utils", utils)
__all__ = getattr(_me, "__all__", [n for n in dir(_me) if not n.startswith("_")])
