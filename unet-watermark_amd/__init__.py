"""unet-watermark_amd — MI355X-native U-Net watermark-segmentation hot path.

Python host layer over libuwm.so (hand-written HIP for gfx950, C ABI in include/uwm.h):
`Unet` / `UnetPlusPlus` (smp drop-ins), Dice/BCE/Combined losses, binary metrics, trainer / predictor.
Import name: `unet_watermark_amd` (alias package next to this directory).
"""
from . import _lib  # noqa: F401
from .model import Unet, UnetPlusPlus, create_model, create_model_from_config, SUPPORTED_MODELS  # noqa: F401
from .losses import DiceLoss, BCEWithLogitsLoss, CombinedLoss, get_loss_function  # noqa: F401
from .metrics import (get_metrics, get_stats, micro_scores, logits_metrics, threshold_mask, resize_threshold,  # noqa: F401
                      dice_coef, iou_score)

__version__ = "0.1.0"

from .data import device_preprocess, aug_flags, random_aug_flags  # noqa: F401,E402
