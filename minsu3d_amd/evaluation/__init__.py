from .instance_segmentation import GeneralDatasetEvaluator, get_gt_instances, rle_decode, rle_encode  # noqa: F401
from .object_detection import evaluate_bbox_acc, get_gt_bbox  # noqa: F401
from .semantic_segmentation import evaluate_semantic_accuracy, evaluate_semantic_miou  # noqa: F401
