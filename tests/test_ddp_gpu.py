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
        losses = []
        for step in range(2):
            x, t = _batch(rank, step)
            losses.append(float(tr.step(x.to(dev), t.to(dev))[0]))
        torch.cuda.synchronize()
        torch.save({"params": m.flat_parameters().cpu(), "losses": losses}, os.path.join(outdir, f"rank{rank}.pt"))
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
    res = [(r, out[r]["params"], out[r]["losses"]) for r in range(2)]
    p0, p1 = res[0][1], res[1][1]
    assert torch.equal(p0, p1)                             # replicas stay bit-identical
    # single-process reference: rank 0's initial weights, per step the mean of the two ranks' gradients, same Adam
    torch.manual_seed(100)
    m = getattr(U, arch)("resnet18").to(cuda)
    opt = FusedAdam(m, lr=1e-3, eps=1e-3)
    crit = U.DiceLoss(mode="binary", smooth=1e-5)
    m.train()
    for step in range(2):
        gs = []
        for r in range(2):
            x, t = _batch(r, step)
            loss = crit(m(x.to(cuda)), t.unsqueeze(1).to(cuda))
            assert abs(float(loss.detach()) - res[r][2][step]) < 2e-5, (step, r, float(loss.detach()), res[r][2][step])
            loss.backward()
            gs.append(m.flat_grads().clone())
        m.flat_grads().copy_(0.5 * (gs[0] + gs[1]))
        opt.step()
    diff = (m.flat_parameters().cpu() - p0).abs().max()
    # a few % of one Adam step (lr 1e-3): summation-order noise of the fp32 weight-gradient atomics through Adam's
    # normalisation (measured 0.7e-5 Unet, 3.4e-5 UnetPlusPlus); a wrong bucket / scale would be >= 1e-3
    assert diff < 1e-4, float(diff)
