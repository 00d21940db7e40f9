"""The CLI's data-parallel loop with EARLY STOPPING ON (/root/reference/src/train.py:37-66,362-368,413-420), two ranks
sharing the one MI355X of this box (gloo carries the collectives; RCCL refuses duplicate devices): every rank must take
the same LR-schedule / best-model / stop decision from ONE all-reduced validation loss, so nobody leaves the gradient
all-reduce early (round 1's loop updated `best` on rank 0 only: rank 0 stopped, the others hung).  Also a 2-rank RCCL
Trainer step, skipped where fewer than two devices are visible."""
import json
import os
import socket

import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _cli_worker(rank, world, port, outdir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank), UWM_DIST_BACKEND="gloo")
    from unet_watermark_amd import cli
    import torch as T
    # a vanishing learning rate: once the BatchNorm running statistics have settled the validation loss only jitters, so
    # patience 2 ends the run long before the last epoch — on EVERY rank, in the same epoch
    hist = cli.main(["train", "--epochs", "60", "--batch-size", "2", "--lr", "1e-9", "--early-stopping-patience", "2",
                     "--synthetic", "16", "--img-size", "64", "--encoder", "resnet18", "--model", "Unet", "--workers", "0",
                     "--optimizer", "SGD", "--lr-scheduler", "CosineAnnealingLR",
                     "--model-save-path", os.path.join(outdir, f"best_rank{rank}.pth"),
                     "--checkpoint-dir", os.path.join(outdir, f"ck_rank{rank}")])
    T.cuda.synchronize()
    json.dump(hist, open(os.path.join(outdir, f"hist{rank}.json"), "w"))


def test_cli_two_ranks_early_stopping_is_rank_consistent(cuda, tmp_path):
    ctx = mp.get_context("spawn")
    port = _free_port()
    procs = [ctx.Process(target=_cli_worker, args=(r, 2, port, str(tmp_path))) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(420)
        assert p.exitcode == 0, "a rank hung or failed (early stopping must be taken by every rank in the same epoch)"
    h = [json.load(open(tmp_path / f"hist{r}.json")) for r in range(2)]
    assert len(h[0]) == len(h[1]) and 2 <= len(h[0]) < 60                    # stopped early, together
    for a, b in zip(h[0], h[1]):
        assert a["val_loss"] == b["val_loss"] and a["train_loss"] == b["train_loss"] and a["lr"] == b["lr"]
    assert os.path.exists(tmp_path / "best_rank0.pth") and not os.path.exists(tmp_path / "best_rank1.pth")   # rank 0 writes


def _rccl_worker(rank, world, port, outdir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch as T
    import torch.distributed as dist
    T.cuda.set_device(rank)
    dev = T.device("cuda", rank)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    try:
        import unet_watermark_amd as U
        from unet_watermark_amd.train import Trainer
        from oracle import unet_oracle as O
        T.manual_seed(100 + rank)
        m = U.Unet("resnet18").to(dev)
        tr = Trainer(m, lr=1e-3, adam_eps=1e-3)
        for step in range(2):
            x, t = O.synthetic_batch(2, 96, 96, seed=1000 + 17 * rank + step)
            tr.step(x.to(dev), t.to(dev))
        T.cuda.synchronize(dev)
        T.save(m.flat_parameters().cpu(), os.path.join(outdir, f"p{rank}.pt"))
    finally:
        dist.destroy_process_group()


def test_two_rank_rccl_step_keeps_replicas_identical(cuda, tmp_path):
    """The real thing — two ranks, two GPUs, RCCL: bucket ordering across ranks, comm stream vs RCCL's own stream."""
    if torch.cuda.device_count() < 2:
        pytest.skip("needs 2 HIP devices (RCCL refuses duplicate devices); the 8-GPU run is the driver's")
    ctx = mp.get_context("spawn")
    port = _free_port()
    procs = [ctx.Process(target=_rccl_worker, args=(r, 2, port, str(tmp_path))) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(300)
        assert p.exitcode == 0
    assert torch.equal(torch.load(tmp_path / "p0.pt"), torch.load(tmp_path / "p1.pt"))
