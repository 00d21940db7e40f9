"""`python main.py train|predict ...` — counterpart of the reference's CLI for the model path
(/root/reference/src/cli.py:368-395 flag names; train loop order /root/reference/src/train.py:207-499:
epochs of train_epoch + validate (val batch = 2x train batch, :254), ReduceLROnPlateau on the val loss
(:280-296,408-412), best / periodic checkpoints (:428-460), early stopping (:37-66,413-420)).
`repair` / `auto` (IOPaint, OCR, video, data synthesis) are out of scope (SURVEY.md §2 rows 6,9,10).
Multi-GPU: launch with torch.distributed.run; every rank trains its shard, gradients are all-reduced."""
from __future__ import annotations

import argparse
import json
import os
import time

import torch
import torch.distributed as dist
from torch.utils.data import DataLoader, Subset
from torch.utils.data.distributed import DistributedSampler

from .checkpoint import load_checkpoint, save_checkpoint
from .config import get_cfg_defaults, update_config
from .data import FolderDataset, SyntheticWatermarkDataset
from .losses import get_loss_function
from .metrics import logits_metrics
from .model import create_model_from_config
from .predict import WatermarkPredictor
from .train import Trainer


def _loss_weights(cfg):
    name = cfg.LOSS.NAME
    if name == "DiceLoss":
        return 1.0, 0.0
    if name == "BCEWithLogitsLoss":
        return 0.0, 1.0
    if name == "CombinedLoss":
        return float(cfg.LOSS.DICE_WEIGHT), float(cfg.LOSS.BCE_WEIGHT)
    raise ValueError(f"unsupported LOSS.NAME {name!r} (DiceLoss | BCEWithLogitsLoss | CombinedLoss)")


def _datasets(cfg, synthetic):
    if synthetic or not os.path.isdir(os.path.join(cfg.DATA.ROOT_DIR, "watermarked")):
        full = SyntheticWatermarkDataset(int(synthetic or 256), cfg.DATA.IMG_SIZE, cfg.DATA.SEED)
    else:
        full = FolderDataset(cfg.DATA.ROOT_DIR, cfg.DATA.IMG_SIZE)
    n = len(full)
    g = torch.Generator().manual_seed(int(cfg.DATA.SEED))
    perm = torch.randperm(n, generator=g).tolist() if cfg.DATA.SHUFFLE else list(range(n))
    ntr = max(1, int(n * float(cfg.DATA.TRAIN_RATIO)))
    return Subset(full, perm[:ntr]), Subset(full, perm[ntr:] or perm[:1])


@torch.no_grad()
def _validate(model, loader, criterion, device):
    model.eval()
    tot, nb, agg = 0.0, 0, {}
    for x, t in loader:
        x, t = x.to(device, non_blocking=True), t.to(device, non_blocking=True)
        out = model(x)
        tot += float(criterion(out, t.unsqueeze(1)))
        for k, v in logits_metrics(out, t).items():
            agg[k] = agg.get(k, 0.0) + v
        nb += 1
    nb = max(nb, 1)
    return tot / nb, {k: v / nb for k, v in agg.items()}


def train_command(args):
    cfg = get_cfg_defaults()
    if args.config and os.path.exists(args.config):
        update_config(cfg, args.config)
    for flag, (sec, key) in dict(data_dir=("DATA", "ROOT_DIR"), output_dir=("TRAIN", "OUTPUT_DIR"),
                                 model_save_path=("TRAIN", "MODEL_SAVE_PATH"), batch_size=("TRAIN", "BATCH_SIZE"),
                                 epochs=("TRAIN", "EPOCHS"), lr=("TRAIN", "LR"), img_size=("DATA", "IMG_SIZE"),
                                 early_stopping_patience=("TRAIN", "EARLY_STOPPING_PATIENCE")).items():
        if getattr(args, flag, None) is not None:
            cfg[sec][key] = getattr(args, flag)
    if args.no_early_stopping:
        cfg.TRAIN.USE_EARLY_STOPPING = False
    if args.encoder:
        cfg.MODEL.ENCODER_NAME = args.encoder
    if getattr(args, "model", None):
        cfg.MODEL.NAME = args.model
    if cfg.MODEL.NAME not in ("Unet", "UnetPlusPlus"):
        raise ValueError(f"MODEL.NAME={cfg.MODEL.NAME!r}: this build serves 'Unet' and 'UnetPlusPlus'")
    if cfg.MODEL.ENCODER_WEIGHTS is not None:
        print(f"note: ENCODER_WEIGHTS={cfg.MODEL.ENCODER_WEIGHTS!r} needs a download; training from seeded init")
        cfg.MODEL.ENCODER_WEIGHTS = None

    world = int(os.environ.get("WORLD_SIZE", "1")); rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("training needs a HIP device (this path has no CPU fallback)")
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=device)
    torch.manual_seed(int(cfg.DATA.SEED))
    model = create_model_from_config(cfg).to(device)
    wd, wb = _loss_weights(cfg)
    trainer = Trainer(model, w_dice=wd, w_bce=wb, smooth=float(cfg.LOSS.SMOOTH), lr=float(cfg.TRAIN.LR),
                      weight_decay=float(cfg.TRAIN.WEIGHT_DECAY),
                      max_grad_norm=(float(cfg.TRAIN.GRADIENT_CLIP) if args.grad_clip else None))
    criterion = get_loss_function(cfg)
    start_epoch, best = 0, float("inf")
    if args.resume:
        ck = load_checkpoint(args.resume, model, trainer.opt)
        start_epoch = int(ck.get("epoch", 0))            # stored epoch is already +1 (Appendix B.9)
        best = ck.get("best_val_loss", ck.get("val_loss")) or best
    sched = torch.optim.lr_scheduler.ReduceLROnPlateau(trainer.opt, mode="min", factor=float(cfg.OPTIMIZER.SCHEDULER_FACTOR),
                                                       patience=int(cfg.OPTIMIZER.SCHEDULER_PATIENCE))
    tr_set, va_set = _datasets(cfg, args.synthetic)
    bs = int(cfg.TRAIN.BATCH_SIZE)
    sampler = DistributedSampler(tr_set, world, rank, shuffle=True, seed=int(cfg.DATA.SEED)) if world > 1 else None
    tr = DataLoader(tr_set, bs, shuffle=sampler is None, sampler=sampler, num_workers=int(args.workers), drop_last=True,
                    pin_memory=True)
    va = DataLoader(va_set, bs * 2, shuffle=False, num_workers=int(args.workers), pin_memory=True)
    hist, bad = [], 0
    for epoch in range(start_epoch, int(cfg.TRAIN.EPOCHS)):
        if sampler is not None:
            sampler.set_epoch(epoch)
        model.train()
        t0, seen = time.time(), 0
        acc = torch.zeros(3, device=device)
        for x, t in tr:
            acc += trainer.step(x.to(device, non_blocking=True), t.to(device, non_blocking=True))
            seen += x.shape[0]
        torch.cuda.synchronize(device)
        dt = time.time() - t0
        tl = float(acc[0]) / max(1, len(tr))
        vl, vm = _validate(model, va, criterion, device)
        sched.step(vl)
        rec = dict(epoch=epoch + 1, train_loss=tl, val_loss=vl, val_metrics=vm, lr=trainer.opt.param_groups[0]["lr"],
                   images_per_sec=world * seen / max(dt, 1e-9))
        hist.append(rec)
        if rank == 0:
            print(json.dumps(rec), flush=True)
            if vl < best:
                best = vl
                save_checkpoint(cfg.TRAIN.MODEL_SAVE_PATH, model, epoch + 1, vl, vm, cfg)
            interval = max(5, int(cfg.TRAIN.EPOCHS) // 10)
            if (epoch + 1) % interval == 0 or epoch >= int(cfg.TRAIN.EPOCHS) - 3:
                save_checkpoint(os.path.join(cfg.TRAIN.CHECKPOINT_DIR, f"checkpoint_epoch_{epoch + 1:03d}.pth"), model,
                                epoch + 1, vl, vm, cfg, optimizer=trainer.opt, scheduler=sched, train_loss=tl,
                                best_val_loss=best, history=hist)
        bad = 0 if vl <= best else bad + 1
        if cfg.TRAIN.USE_EARLY_STOPPING and bad >= int(cfg.TRAIN.EARLY_STOPPING_PATIENCE):
            break
    if world > 1:
        dist.destroy_process_group()
    return hist


def predict_command(args):
    cfg = get_cfg_defaults()
    if args.config and os.path.exists(args.config):
        update_config(cfg, args.config)
    if args.encoder:
        cfg.MODEL.ENCODER_NAME = args.encoder
    if args.threshold is not None:
        cfg.PREDICT.THRESHOLD = args.threshold
    pred = WatermarkPredictor(args.model, None, cfg, device="cuda")
    from PIL import Image
    import numpy as np
    os.makedirs(args.output, exist_ok=True)
    files = sorted(f for f in os.listdir(args.input) if f.lower().endswith((".png", ".jpg", ".jpeg")))
    s, bs = int(cfg.DATA.IMG_SIZE), int(args.batch_size or cfg.PREDICT.BATCH_SIZE)
    for i in range(0, len(files), bs):
        chunk = files[i:i + bs]
        ims = [Image.open(os.path.join(args.input, f)).convert("RGB") for f in chunk]
        arr = np.stack([np.asarray(im.resize((s, s), Image.BILINEAR), dtype=np.uint8) for im in ims])
        logits = pred.logits(pred.preprocess(torch.from_numpy(arr)), use_graph=len(chunk) == bs)
        from .metrics import resize_threshold
        for k, (f, im) in enumerate(zip(chunk, ims)):          # bilinear resize of the raw logits to the original size, then threshold
            m = resize_threshold(logits[k:k + 1], (im.size[1], im.size[0]), pred.threshold, args.sigmoid)[0].cpu().numpy()
            Image.fromarray(m).save(os.path.join(args.output, os.path.splitext(f)[0] + "_mask.png"))
    print(f"wrote {len(files)} masks to {args.output}")


def main(argv=None):
    ap = argparse.ArgumentParser(description="MI355X-native U-Net watermark segmentation (train | predict)")
    sub = ap.add_subparsers(dest="command")
    tp = sub.add_parser("train")
    tp.add_argument("--config", type=str, default=None)
    tp.add_argument("--device", type=str, default="auto")
    tp.add_argument("--data-dir", type=str); tp.add_argument("--output-dir", type=str)
    tp.add_argument("--model-save-path", type=str); tp.add_argument("--batch-size", type=int)
    tp.add_argument("--epochs", type=int); tp.add_argument("--lr", type=float)
    tp.add_argument("--no-early-stopping", action="store_true")
    tp.add_argument("--early-stopping-patience", type=int)
    tp.add_argument("--resume", type=str)
    tp.add_argument("--use-blurred-mask", action="store_true", help="accepted for compatibility (dataset-side option)")
    tp.add_argument("--encoder", type=str); tp.add_argument("--img-size", type=int)
    tp.add_argument("--synthetic", type=int, default=0, help="train on N synthetic images instead of DATA.ROOT_DIR")
    tp.add_argument("--workers", type=int, default=2)
    tp.add_argument("--model", choices=["Unet", "UnetPlusPlus"], default=None, help="MODEL.NAME (reference default: UnetPlusPlus)")
    tp.add_argument("--grad-clip", action="store_true", help="honour TRAIN.GRADIENT_CLIP (the reference defines but never applies it)")
    pp = sub.add_parser("predict")
    pp.add_argument("--input", type=str, required=True); pp.add_argument("--output", type=str, required=True)
    pp.add_argument("--model", type=str, required=True); pp.add_argument("--config", type=str, default=None)
    pp.add_argument("--encoder", type=str); pp.add_argument("--threshold", type=float)
    pp.add_argument("--batch-size", type=int); pp.add_argument("--sigmoid", action="store_true")
    args = ap.parse_args(argv)
    if args.command == "train":
        return train_command(args)
    if args.command == "predict":
        return predict_command(args)
    ap.print_help()
