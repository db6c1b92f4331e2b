from .seg_unet import SegUNet_F  # noqa: F401
