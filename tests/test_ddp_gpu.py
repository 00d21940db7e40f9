"""Two data-parallel ranks on ONE MI355X (gloo carries the collectives so both ranks may share the device; RCCL
refuses duplicate GPUs): the full GPU training step of SURVEY.md 8(e) — initial broadcast, rank-distinct batches,
staged backward with the bucketed all-reduce on the communication stream, fused Adam with grad_scale = 1/world —
must leave both ranks with bit-identical parameters equal to one process applying the AVERAGE of the two ranks'
gradients."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _batch(rank, step, n=2, s=96):
    from oracle import unet_oracle as O
    return O.synthetic_batch(n, s, s, seed=1000 + 17 * rank + step)


def _worker(rank, world, port, arch, outdir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import unet_watermark_amd as U
        from unet_watermark_amd.train import Trainer
        dev = torch.device("cuda:0")
        torch.manual_seed(100 + rank)                      # different initial weights: the Trainer must broadcast rank 0's
        m = getattr(U, arch)("resnet18").to(dev)
        tr = Trainer(m, w_dice=1.0, w_bce=0.0, smooth=1e-5, lr=1e-3, adam_eps=1e-3)
        x, t = _batch(rank, 0)
        loss0 = float(tr.step(x.to(dev), t.to(dev))[0])
        torch.cuda.synchronize()
        out = {"params1": m.flat_parameters().cpu().clone(), "gsum": m.flat_grads().cpu().clone(), "loss0": loss0}
        x, t = _batch(rank, 1)
        out["loss1"] = float(tr.step(x.to(dev), t.to(dev))[0])
        torch.cuda.synchronize()
        out["params2"] = m.flat_parameters().cpu().clone()
        torch.save(out, os.path.join(outdir, f"rank{rank}.pt"))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("arch", ["Unet", "UnetPlusPlus"])
def test_two_ranks_one_gpu_match_averaged_gradient_step(cuda, arch, tmp_path):
    import unet_watermark_amd as U
    from unet_watermark_amd.train import FusedAdam
    ctx = mp.get_context("spawn")
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, arch, str(tmp_path))) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(300)
        assert p.exitcode == 0
    out = [torch.load(tmp_path / f"rank{r}.pt") for r in range(2)]
    # replicas stay bit-identical through two steps; the all-reduced gradient arenas are identical too
    assert torch.equal(out[0]["params1"], out[1]["params1"]) and torch.equal(out[0]["params2"], out[1]["params2"])
    assert torch.equal(out[0]["gsum"], out[1]["gsum"])
    # single-process reference from rank 0's initial weights
    torch.manual_seed(100)
    m = getattr(U, arch)("resnet18").to(cuda)
    opt = FusedAdam(m, lr=1e-3, eps=1e-3)
    crit = U.DiceLoss(mode="binary", smooth=1e-5)
    m.train()
    gs = []
    for r in range(2):
        x, t = _batch(r, 0)
        loss = crit(m(x.to(cuda)), t.unsqueeze(1).to(cuda))
        assert abs(float(loss.detach()) - out[r]["loss0"]) < 2e-5, (r, float(loss.detach()), out[r]["loss0"])   # broadcast worked
        loss.backward()
        gs.append(m.flat_grads().clone())
    # (1) every bucket was summed over the ranks (only summation-order noise of the fp32 weight-gradient atomics)
    ref_sum = (gs[0] + gs[1]).cpu().double(); got = out[0]["gsum"].double()
    assert float((got - ref_sum).norm() / ref_sum.norm()) < 1e-4
    assert float((got - ref_sum).abs().max()) < 1e-4 * float(ref_sum.abs().max()) + 1e-7
    # (2) the optimizer applied the MEAN: the same Adam kernel on the ranks' summed gradient with grad_scale 1/2
    m.flat_grads().copy_(out[0]["gsum"].to(cuda))
    opt.step(grad_scale=0.5)
    assert float((m.flat_parameters().cpu() - out[0]["params1"]).abs().max()) < 1e-7
    # (3) second step: per-rank losses of the updated replicas match this process's
    for r in range(2):
        x, t = _batch(r, 1)
        with torch.no_grad():
            l1 = crit(m(x.to(cuda)), t.unsqueeze(1).to(cuda))
        assert abs(float(l1) - out[r]["loss1"]) < 2e-5, (r, float(l1), out[r]["loss1"])


def _global_dice_worker(rank, world, port, outdir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import unet_watermark_amd as U
        from unet_watermark_amd.train import Trainer
        dev = torch.device("cuda:0")
        torch.manual_seed(100)
        m = U.Unet("resnet18").to(dev)
        tr = Trainer(m, w_dice=0.7, w_bce=0.3, smooth=1.0, lr=1e-3, adam_eps=1e-3, global_dice=True)
        x, t = _batch(rank, 0)
        loss = tr.step(x.to(dev), t.to(dev)).cpu().clone()
        torch.cuda.synchronize()
        torch.save({"loss": loss, "gsum": m.flat_grads().cpu().clone()}, os.path.join(outdir, f"gd{rank}.pt"))
    finally:
        dist.destroy_process_group()


def test_global_dice_two_ranks_equal_one_process_on_the_concatenated_batch(cuda, tmp_path):
    """SURVEY.md 8(e) caveat: with `global_dice` the ranks all-reduce the four loss sums and back-propagate the Dice (+BCE) of
    the global batch.  Eval-mode BatchNorm would be needed for an exact one-process twin on rank-distinct data (batch
    statistics stay per rank, as under torch DDP), so the twin here is built the other way round: one process, train mode,
    on the concatenation of the two rank batches, gradients taken THROUGH the per-rank statistics — i.e. the same two
    forward/backward passes, only the loss coupled through the summed {sum p*t, sum p, sum t, sum bce}."""
    import unet_watermark_amd as U
    ctx = mp.get_context("spawn")
    port = _free_port()
    procs = [ctx.Process(target=_global_dice_worker, args=(r, 2, port, str(tmp_path))) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(300)
        assert p.exitcode == 0
    out = [torch.load(tmp_path / f"gd{r}.pt") for r in range(2)]
    assert torch.equal(out[0]["loss"], out[1]["loss"])             # every rank reports the GLOBAL loss
    assert torch.equal(out[0]["gsum"], out[1]["gsum"])
    # one-process twin: per-rank forward (own BatchNorm statistics), global loss from the summed sums, per-rank backward
    torch.manual_seed(100)
    m = U.Unet("resnet18").to(cuda)
    m.train()
    xs, ts, logits = [], [], []
    for r in range(2):
        x, t = _batch(r, 0)
        xs.append(x.to(cuda)); ts.append(t.to(cuda).float())
    smooth, wd, wb = 1.0, 0.7, 0.3
    with torch.no_grad():
        for r in range(2):
            logits.append(m(xs[r])[:, 0].clone())
    z = torch.cat(logits).double().requires_grad_(True)
    tt = torch.cat(ts).double()
    p = torch.sigmoid(z)
    dice = 1.0 - (2.0 * (p * tt).sum() + smooth) / ((p + tt).sum() + smooth).clamp_min(1e-7)
    bce = torch.nn.functional.binary_cross_entropy_with_logits(z, tt)
    loss = wd * dice + wb * bce
    loss.backward()
    assert abs(float(loss.detach()) - float(out[0]["loss"][0])) < 2e-5 and abs(float(dice.detach()) - float(out[0]["loss"][1])) < 2e-5
    n = xs[0].shape[0]
    gsum = torch.zeros_like(m.flat_grads())
    for r in range(2):
        y = m(xs[r])                                             # rebuilds rank r's training workspace (running stats drift is irrelevant here)
        y.backward(z.grad[r * n:(r + 1) * n].float().unsqueeze(1))
        gsum += m.flat_grads()
    # the ranks' arenas hold world x (g0 + g1) before the 1/world of the optimizer: dlogits carried grad_scale = world
    got = out[0]["gsum"].double() / 2.0; ref = gsum.cpu().double()
    assert float((got - ref).norm() / ref.norm()) < 1e-4
