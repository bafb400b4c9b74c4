from .general_model import GeneralModel, clusters_voxelization, get_segmented_scores  # noqa: F401
from .pointgroup import PointGroup  # noqa: F401
from .hais import HAIS  # noqa: F401
from .softgroup import SoftGroup  # noqa: F401
