"""Configuration: the reference's yacs tree (/root/reference/src/configs/config.py:8-96) restated with
PyYAML + attribute nodes (yacs is not in this image).  Same keys, same YAML files; keys the reference
defines but never reads are accepted, and the loss weights are honoured.
Defaults are the reference's (MODEL.NAME = "UnetPlusPlus", /root/reference/src/configs/config.py:15) with one
exception, because this build has no network: MODEL.ENCODER_WEIGHTS = None (reference default "imagenet", a
download).  BASELINE.json's configs name "Unet": bench.py and the tests pass the architecture explicitly."""
from __future__ import annotations

import copy

import yaml


class CfgNode(dict):
    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = v

    def clone(self):
        return copy.deepcopy(self)

    def to_dict(self):
        return {k: (v.to_dict() if isinstance(v, CfgNode) else v) for k, v in self.items()}


def _node(d):
    return CfgNode({k: (_node(v) if isinstance(v, dict) else v) for k, v in d.items()})


_DEFAULTS = {
    "DEVICE": "cuda",
    "MODEL": dict(NAME="UnetPlusPlus", ENCODER_NAME="resnet34", ENCODER_WEIGHTS=None, ENCODER_DEPTH=5,
                  DECODER_CHANNELS=[256, 128, 64, 32, 16], IN_CHANNELS=3, CLASSES=1, ACTIVATION=None),
    "DATA": dict(ROOT_DIR="data/train", ADDITIONAL_ROOT_DIRS=[], IMG_SIZE=512, GENERATE_MASK_THRESHOLD=30,
                 TRAIN_RATIO=0.8, VAL_RATIO=0.2, SHUFFLE=True, SEED=42, NUM_WORKERS=4, CACHE_IMAGES=False,
                 PREFETCH_FACTOR=2, AUGMENTATION_TYPE="transparent_watermark"),
    "TRAIN": dict(BATCH_SIZE=16, EPOCHS=300, LR=1e-4, WEIGHT_DECAY=1e-4, OUTPUT_DIR="logs/output",
                  MODEL_SAVE_PATH="models/unet_watermark.pth", LOG_INTERVAL=10, SAVE_INTERVAL=50,
                  USE_EARLY_STOPPING=True, EARLY_STOPPING_PATIENCE=10, CHECKPOINT_DIR="models/checkpoints",
                  SAVE_BEST_ONLY=False, USE_AMP=False, GRADIENT_CLIP=1.0),
    "LOSS": dict(NAME="DiceLoss", MODE="binary", SMOOTH=1e-5, BCE_WEIGHT=0.5, DICE_WEIGHT=0.5, DICE_SMOOTH=1e-5,
                 FOCAL_ALPHA=0.25, FOCAL_GAMMA=2.0),
    "OPTIMIZER": dict(NAME="Adam", LR_SCHEDULER="ReduceLROnPlateau", SCHEDULER_PATIENCE=5, SCHEDULER_FACTOR=0.5),
    "PREDICT": dict(INPUT_PATH="data/input", OUTPUT_DIR="data/output", BATCH_SIZE=8, AUTO_BATCH_SIZE=True,
                    MAX_BATCH_SIZE=32, THRESHOLD=0.5, POST_PROCESS=True),
    "VAL": dict(METRICS=["dice", "iou", "accuracy"]),
}


def get_cfg_defaults() -> CfgNode:
    return _node(copy.deepcopy(_DEFAULTS))


def _merge(dst: CfgNode, src: dict, path=""):
    for k, v in src.items():
        if isinstance(v, dict):
            if k not in dst or not isinstance(dst[k], CfgNode):
                dst[k] = CfgNode()
            _merge(dst[k], v, path + k + ".")
        else:
            dst[k] = v           # unknown leaf keys are accepted (yacs would reject unet_text_watermark.yaml)


def update_config(cfg: CfgNode, config_file: str) -> CfgNode:
    with open(config_file, "r", encoding="utf-8") as f:
        data = yaml.safe_load(f) or {}
    _merge(cfg, data)
    return cfg
