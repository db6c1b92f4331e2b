from .seg_unet import SegUNet_F  # noqa: F401
from .sr_loss import LazyScalars, RecLoss, SRLoss  # noqa: F401
