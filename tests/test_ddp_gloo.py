"""gloo tests (CPU, world_size 2 AND 8) of the data-parallel exchange: the bucketed gradient all-reduce over the
flat arena (GradReducer), the initial parameter broadcast, and the self-validation bench.py runs before it times an
N-GPU step (bench.ddp_verdict: checksums across ranks, comparison with one plain all-reduce, the all-reduced verdict
that makes EVERY rank exit non-zero on a mismatch) — the host logic of SURVEY.md §8(e).  The collective backend on the
GPUs is RCCL ("nccl"); the bucket / ordering / verdict logic is backend-independent, and world 8 is the size the
driver's scaling run uses (no 8-GPU box is available to the build, so the logic runs here)."""
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


@pytest.mark.parametrize("world", [2, 8])
def test_bucketed_allreduce_and_broadcast_gloo(world):
    import __graft_entry__ as g
    g.build()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert sorted(r[0] for r in res) == list(range(world))
    for rank, same_params, moved, ok_sum, covered in res:
        assert same_params and moved and ok_sum and covered, (rank, same_params, moved, ok_sum, covered)


def _verdict_worker(rank, world, port, q, corrupt):
    """bench.ddp_verdict over gloo with fabricated arenas: 5 buckets like the model's backward stages, rank-distinct local gradients,
    the step under test = the real GradReducer (issued in backward order) + a toy optimizer step.  corrupt = (rank, what): that rank
    breaks the exchange after the fact — the verdict must make EVERY rank exit 3."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    from unet_watermark_amd.train import GradReducer
    n = 100003
    cuts = [0, 31001, 52000, 52000, 80001, n]               # (one empty bucket, as a model without parameters in a stage has)
    stages = [(cuts[i], cuts[i + 1]) for i in range(5)]
    g = torch.Generator().manual_seed(7 + (0 if corrupt and corrupt[1] == "same_data" else rank))
    local = torch.randn(n, generator=g) * 1e-3
    params0 = torch.randn(n, generator=torch.Generator().manual_seed(1))

    def real_step():
        grads = local.clone()
        red = GradReducer(grads, stages)
        for k in range(len(red.buckets)):
            red.reduce(k)
        red.finish()
        if corrupt and corrupt[0] == rank and corrupt[1] == "bits":
            grads[40000] += 1e-7                                 # one element of bucket 1, one ulp-scale nudge on ONE rank
        params = params0 - 0.1 * grads / world
        if corrupt and corrupt[0] == rank and corrupt[1] == "params":
            params[5] += 1.0
        return grads, params

    rows, lvals, psame = bench.ddp_verdict(local, real_step, stages, world, rank, torch.device("cpu"))
    q.put((rank, len(rows), all(r["bit_identical_across_ranks"] for r in rows), psame, len(set(lvals))))
    dist.destroy_process_group()


@pytest.mark.parametrize("corrupt,expect_exit", [(None, 0), ((3, "bits"), 3), ((5, "params"), 3), ((0, "same_data"), 3)])
def test_bench_ddp_verdict_world8(corrupt, expect_exit):
    world = 8
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_verdict_worker, args=(r, world, port, q, corrupt)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(300)
    assert [p.exitcode for p in procs] == [expect_exit] * world, [p.exitcode for p in procs]      # EVERY rank agrees on the verdict
    if expect_exit == 0:
        res = [q.get(timeout=10) for _ in procs]
        for rank, nrows, same, psame, distinct in res:
            assert nrows == 4 and same and psame and distinct == world, (rank, nrows, same, psame, distinct)
