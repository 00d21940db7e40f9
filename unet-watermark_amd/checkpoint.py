"""Checkpoint I/O in the reference's dict format (/root/reference/src/train.py:428-458): best =
{epoch, model_state_dict, val_loss, val_metrics, config}; periodic adds optimizer/scheduler state and the
loss/metric histories.  `config` is stored as a plain dict (the reference pickles a yacs CfgNode, which
needs weights_only=False — Appendix B.12); both that and a bare state_dict (old format,
/root/reference/src/predict.py:88-91) are read."""
from __future__ import annotations

import os

import torch


def _cfg_dict(cfg):
    if cfg is None:
        return None
    return cfg.to_dict() if hasattr(cfg, "to_dict") else dict(cfg)


def save_checkpoint(path, model, epoch, val_loss=None, val_metrics=None, cfg=None, optimizer=None, scheduler=None,
                    **extra):
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    sd = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}     # real copies (Appendix B.4)
    info = {"epoch": int(epoch), "model_state_dict": sd, "val_loss": val_loss, "val_metrics": val_metrics,
            "config": _cfg_dict(cfg)}
    if optimizer is not None:
        osd = optimizer.state_dict()
        info["optimizer_state_dict"] = {k: (v.detach().cpu() if torch.is_tensor(v) else v) for k, v in osd.items()}
    if scheduler is not None:
        info["scheduler_state_dict"] = scheduler.state_dict()
    info.update(extra)
    torch.save(info, path)
    return path


def load_checkpoint(path, model=None, optimizer=None, map_location="cpu"):
    """Returns the checkpoint dict (a bare state_dict is wrapped).  Loads model/optimizer when given."""
    try:
        ck = torch.load(path, map_location=map_location, weights_only=True)
    except Exception:
        ck = torch.load(path, map_location=map_location, weights_only=False)     # reference checkpoints pickle a CfgNode
    if not (isinstance(ck, dict) and "model_state_dict" in ck):
        ck = {"epoch": 0, "model_state_dict": ck, "val_loss": None, "val_metrics": None, "config": None}
    if model is not None:
        model.load_state_dict(ck["model_state_dict"])
    if optimizer is not None and ck.get("optimizer_state_dict") is not None:
        optimizer.load_state_dict(ck["optimizer_state_dict"])
    return ck
