from .pt_offset_loss import PTOffsetLoss  # noqa: F401
