"""world_size-2 gloo test (CPU) of the data-parallel exchange: the bucketed gradient all-reduce over the
flat arena (GradReducer) and the initial parameter broadcast — the host logic of SURVEY.md §8(e).
The collective backend on the GPUs is RCCL ("nccl"); the bucket/ordering logic is backend-independent."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import unet_watermark_amd as U
        from unet_watermark_amd.train import GradReducer, broadcast_model
        torch.manual_seed(100 + rank)                      # ranks start from DIFFERENT weights
        m = U.Unet("resnet18")
        before = m.flat_parameters().clone()
        broadcast_model(m, src=0)
        ref = [torch.zeros_like(before) for _ in range(world)]
        dist.all_gather(ref, m.flat_parameters())
        same_params = all(torch.equal(r, ref[0]) for r in ref)
        moved = (rank == 0) or (not torch.equal(before, m.flat_parameters()))
        # fake per-rank gradients in a flat arena with the model's bucket ranges
        n = m.flat_parameters().numel()
        grads = torch.arange(n, dtype=torch.float32) * 1e-6 + (rank + 1)
        red = GradReducer(grads, m.stages)
        for k in reversed(range(len(red.buckets))):        # any issue order must work
            red.reduce(k)
        red.finish()
        expect = torch.arange(n, dtype=torch.float32) * 1e-6 * world + sum(r + 1 for r in range(world))
        ok_sum = torch.allclose(grads, expect, rtol=1e-6)
        covered = sum(e - b for b, e in red.buckets) == n
        q.put((rank, same_params, moved, ok_sum, covered))
    finally:
        dist.destroy_process_group()


def test_bucketed_allreduce_and_broadcast_gloo_world2():
    import __graft_entry__ as g
    g.build()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for rank, same_params, moved, ok_sum, covered in res:
        assert same_params and moved and ok_sum and covered, (rank, same_params, moved, ok_sum, covered)
