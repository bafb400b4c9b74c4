"""The MinkowskiEngine subset the reference models import (`import MinkowskiEngine as ME`), served by the
gfx950 sparse-voxel engine.  Exactly the 9 symbols the reference uses:

    SparseTensor, MinkowskiConvolution, MinkowskiConvolutionTranspose, MinkowskiBatchNorm, MinkowskiReLU,
    cat, utils.sparse_quantize, utils.sparse_collate  (+ tensor attributes .features/.F, .coordinates/.C)

Module/parameter names match ME so reference state_dicts keep their keys (`kernel`, `bn.weight`, ...).
"""
from . import utils  # noqa: F401
from .tensor import SparseTensor, CoordinateManager, cat, prefetch_coordinates  # noqa: F401
from .modules import (MinkowskiConvolution, MinkowskiConvolutionTranspose, MinkowskiBatchNorm,  # noqa: F401
                      MinkowskiReLU, prepare_conv_weights, release_conv_weights)
from .functional import gather_rows  # noqa: F401  (engine extra: x[idx] with a scatter-add backward)
from .functional import SkipLink  # noqa: F401  (engine extra: a residual block's skip gradient, see functional.py)
