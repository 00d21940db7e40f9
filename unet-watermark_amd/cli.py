"""`python main.py train|predict ...` — counterpart of the reference's CLI for the model path
(/root/reference/src/cli.py:368-395 flag names; train loop order /root/reference/src/train.py:207-499:
epochs of train_epoch + validate (val batch = 2x train batch, :254), Adam | SGD (:265-279), ReduceLROnPlateau on the
val loss | CosineAnnealingLR (:280-296,408-412), best / periodic checkpoints (:428-460), early stopping (:37-66,362-368)).
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


class EarlyStopping:
    """/root/reference/src/train.py:37-66 (patience, min_delta, restore_best_weights) with the shallow-copy quirk fixed
    (real clones).  Every rank holds one and feeds it the SAME (all-reduced) validation loss, so the
    stop decision is rank-consistent by construction."""

    def __init__(self, patience=7, min_delta=0.0, restore_best_weights=True):
        self.patience, self.min_delta, self.restore = int(patience), float(min_delta), restore_best_weights
        self.best_loss, self.counter, self.best_weights = None, 0, None

    def __call__(self, val_loss, model) -> bool:
        if self.best_loss is None or val_loss < self.best_loss - self.min_delta:
            if self.best_loss is not None:
                self.counter = 0
            self.best_loss = val_loss
            if self.restore:
                self.best_weights = {k: v.detach().clone() for k, v in model.state_dict().items()}
        else:
            self.counter += 1
        if self.counter >= self.patience:
            if self.restore and self.best_weights is not None:
                model.load_state_dict(self.best_weights)
            return True
        return False


def _rank_mean(value: float, device, world: int) -> float:
    """Mean over ranks of a host scalar (each rank validates with its own BatchNorm running statistics — the reference
    has no SyncBN — so per-rank validation losses differ slightly; every decision below uses this ONE number)."""
    if world <= 1:
        return float(value)
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item()) / world


def _make_scheduler(cfg, opt):
    """/root/reference/src/train.py:280-296"""
    name = cfg.OPTIMIZER.LR_SCHEDULER
    if name == "ReduceLROnPlateau":
        return torch.optim.lr_scheduler.ReduceLROnPlateau(opt, mode="min", factor=float(cfg.OPTIMIZER.SCHEDULER_FACTOR),
                                                          patience=int(cfg.OPTIMIZER.SCHEDULER_PATIENCE))
    if name == "CosineAnnealingLR":
        return torch.optim.lr_scheduler.CosineAnnealingLR(opt, T_max=int(cfg.TRAIN.EPOCHS))
    return None


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
    if getattr(args, "optimizer", None):
        cfg.OPTIMIZER.NAME = args.optimizer
    if getattr(args, "lr_scheduler", None):
        cfg.OPTIMIZER.LR_SCHEDULER = args.lr_scheduler
    if getattr(args, "checkpoint_dir", None):
        cfg.TRAIN.CHECKPOINT_DIR = args.checkpoint_dir
    if cfg.MODEL.NAME not in ("Unet", "UnetPlusPlus"):
        raise ValueError(f"MODEL.NAME={cfg.MODEL.NAME!r}: this build serves 'Unet' and 'UnetPlusPlus'")
    if cfg.OPTIMIZER.NAME not in ("Adam", "SGD"):
        raise ValueError(f"unsupported optimizer: {cfg.OPTIMIZER.NAME}")           # /root/reference/src/train.py:279
    if cfg.MODEL.ENCODER_WEIGHTS is not None:
        print(f"note: ENCODER_WEIGHTS={cfg.MODEL.ENCODER_WEIGHTS!r} needs a download; training from seeded init")
        cfg.MODEL.ENCODER_WEIGHTS = None

    world = int(os.environ.get("WORLD_SIZE", "1")); rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("training needs a HIP device (this path has no CPU fallback)")
    ndev = torch.cuda.device_count()
    backend = os.environ.get("UWM_DIST_BACKEND", "nccl")      # tests: "gloo" lets several ranks share one GPU (RCCL refuses that)
    if backend == "gloo":
        local = local % max(1, ndev)
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    own_group = False
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend, **({"device_id": device} if backend == "nccl" else {}))
        own_group = True
    torch.manual_seed(int(cfg.DATA.SEED))
    model = create_model_from_config(cfg).to(device)
    wd, wb = _loss_weights(cfg)
    trainer = Trainer(model, w_dice=wd, w_bce=wb, smooth=float(cfg.LOSS.SMOOTH), lr=float(cfg.TRAIN.LR),
                      weight_decay=float(cfg.TRAIN.WEIGHT_DECAY), optimizer=cfg.OPTIMIZER.NAME,
                      max_grad_norm=(float(cfg.TRAIN.GRADIENT_CLIP) if args.grad_clip else None),
                      global_dice=bool(getattr(args, "global_dice", False)))
    criterion = get_loss_function(cfg)
    sched = _make_scheduler(cfg, trainer.opt)
    start_epoch, best = 0, float("inf")
    tr_losses, va_losses, tr_hist, va_hist = [], [], [], []
    if args.resume:
        ck = load_checkpoint(args.resume, model, trainer.opt)
        if sched is not None and ck.get("scheduler_state_dict"):
            sched.load_state_dict(ck["scheduler_state_dict"])
        start_epoch = int(ck.get("epoch", 0))            # stored epoch is already +1 (Appendix B.9)
        b = ck.get("best_val_loss", ck.get("val_loss"))
        best = float(b) if b is not None else best
        tr_losses, va_losses = list(ck.get("train_losses", [])), list(ck.get("val_losses", []))
        tr_hist, va_hist = list(ck.get("train_metrics_history", [])), list(ck.get("val_metrics_history", []))
        if world > 1:                                    # every rank read the file; make the replicas bit-identical anyway
            from .train import broadcast_model
            broadcast_model(model, 0)
    stopper = (EarlyStopping(patience=int(cfg.TRAIN.EARLY_STOPPING_PATIENCE), restore_best_weights=True)
               if cfg.TRAIN.USE_EARLY_STOPPING else None)       # /root/reference/src/train.py:362-368
    tr_set, va_set = _datasets(cfg, args.synthetic)
    bs = int(cfg.TRAIN.BATCH_SIZE)
    sampler = DistributedSampler(tr_set, world, rank, shuffle=True, seed=int(cfg.DATA.SEED)) if world > 1 else None
    tr = DataLoader(tr_set, bs, shuffle=sampler is None, sampler=sampler, num_workers=int(args.workers), drop_last=True,
                    pin_memory=True)
    va = DataLoader(va_set, bs * 2, shuffle=False, num_workers=int(args.workers), pin_memory=True)
    hist = []
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
        tl = _rank_mean(float(acc[0]) / max(1, len(tr)), device, world)
        vl_local, vm = _validate(model, va, criterion, device)
        # ONE validation loss for every rank: LR schedule, best-model bookkeeping and early stopping all read it, so
        # the replicas take the same decisions and nobody leaves the collective early
        vl = _rank_mean(vl_local, device, world)
        vm = {k: _rank_mean(v, device, world) for k, v in sorted(vm.items())}
        if sched is not None:
            if cfg.OPTIMIZER.LR_SCHEDULER == "ReduceLROnPlateau":
                sched.step(vl)
            else:
                sched.step()
        tr_losses.append(tl); va_losses.append(vl); tr_hist.append({}); va_hist.append(vm)
        rec = dict(epoch=epoch + 1, train_loss=tl, val_loss=vl, val_metrics=vm, lr=trainer.opt.param_groups[0]["lr"],
                   images_per_sec=world * seen / max(dt, 1e-9))
        hist.append(rec)
        improved = vl < best
        if improved:
            best = vl                                       # on EVERY rank
        if rank == 0:                                       # rank 0's BatchNorm buffers are the ones saved
            print(json.dumps(rec), flush=True)
            if improved:
                save_checkpoint(cfg.TRAIN.MODEL_SAVE_PATH, model, epoch + 1, vl, vm, cfg)
            interval = max(5, int(cfg.TRAIN.EPOCHS) // 10)
            if (epoch + 1) % interval == 0 or epoch >= int(cfg.TRAIN.EPOCHS) - 3:
                save_checkpoint(os.path.join(cfg.TRAIN.CHECKPOINT_DIR, f"checkpoint_epoch_{epoch + 1:03d}.pth"), model,
                                epoch + 1, vl, vm, cfg, optimizer=trainer.opt, scheduler=sched, train_loss=tl,
                                train_metrics={}, best_val_loss=best, train_losses=tr_losses, val_losses=va_losses,
                                train_metrics_history=tr_hist, val_metrics_history=va_hist)
        if stopper is not None and stopper(vl, model):
            if rank == 0:
                print(json.dumps({"early_stop": epoch + 1, "best_val_loss": stopper.best_loss}), flush=True)
            break
    # final model (/root/reference/src/train.py:467-485): weights as they stand after the loop — i.e. the RESTORED best weights
    # after an early stop — with the optimizer / scheduler state and is_final=True
    if rank == 0 and hist:
        last = hist[-1]
        save_checkpoint(os.path.join(cfg.TRAIN.CHECKPOINT_DIR, f"final_model_epoch_{last['epoch']:03d}.pth"), model,
                        last["epoch"], last["val_loss"], last["val_metrics"], cfg, optimizer=trainer.opt, scheduler=sched,
                        train_loss=last["train_loss"], train_metrics={}, best_val_loss=best, is_final=True)
    if own_group:
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
    tp.add_argument("--optimizer", choices=["Adam", "SGD"], default=None, help="OPTIMIZER.NAME")
    tp.add_argument("--global-dice", action="store_true",
                    help="data-parallel runs: Dice of the GLOBAL batch (loss sums all-reduced) instead of the mean of per-rank Dice losses")
    tp.add_argument("--lr-scheduler", choices=["ReduceLROnPlateau", "CosineAnnealingLR", "none"], default=None, help="OPTIMIZER.LR_SCHEDULER")
    tp.add_argument("--checkpoint-dir", type=str, default=None, help="TRAIN.CHECKPOINT_DIR")
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
