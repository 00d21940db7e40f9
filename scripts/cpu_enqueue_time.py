"""Host enqueue time per train step (one thread issues every launch of a step): the launch-bound risk of 8 ranks on a 16-core grant
(VERDICT r03 item 8).  Enqueue time = wall time until the last launch of a step has been ISSUED (no sync), measured
after the queue has drained, next to the GPU time per step.

    python scripts/cpu_enqueue_time.py                  one process, unpinned: Unet-resnet34 512^2 bs16 and Unet-efficientnet-b4 1024^2 bs4
    python scripts/cpu_enqueue_time.py --procs 6        P concurrent processes (the box admits 6 on one card), each pinned to 2 cores of its
                                                        own — what a rank of the 8-GPU run gets (16 granted cores / 8 ranks); the processes
                                                        share ONE GPU here, so only the HOST column means anything: the GPU column is P
                                                        steps' worth of work interleaved
"""
import argparse, os, statistics, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def measure(configs, reps, tag, precision):
    import torch
    import unet_watermark_amd as U
    from unet_watermark_amd.train import Trainer
    dev = torch.device("cuda:0")
    for enc, n, s in configs:
        m = U.Unet(enc).to(dev)
        m.set_precision(precision)
        tr = Trainer(m, w_dice=1.0, w_bce=0.0, smooth=1e-5, lr=1e-4, weight_decay=1e-4)
        x = torch.randn(n, 3, s, s, device=dev); t = torch.zeros(n, s, s, dtype=torch.int64, device=dev); t[:, 100:200, 100:300] = 1
        for _ in range(3): tr.step(x, t)
        torch.cuda.synchronize()
        host, gpu = [], []
        for _ in range(reps):
            t0 = time.perf_counter()
            tr.step(x, t)
            t1 = time.perf_counter()
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            host.append(1e3 * (t1 - t0)); gpu.append(1e3 * (t2 - t0))
        print(f"{tag}Unet-{enc} {s}x{s} bs{n} [{precision}]: host enqueue median {statistics.median(host):.2f} ms/step (min {min(host):.2f}, max {max(host):.2f}), "
              f"GPU-complete median {statistics.median(gpu):.2f} ms/step over {reps} single steps", flush=True)
        del m, tr, x, t
        torch.cuda.empty_cache()


def worker(rank, cores_per, reps, precision, barrier):
    cpus = sorted(os.sched_getaffinity(0))
    mine = cpus[(rank * cores_per) % len(cpus): (rank * cores_per) % len(cpus) + cores_per] or cpus[:cores_per]
    os.sched_setaffinity(0, set(mine))
    import torch
    torch.set_num_threads(1)
    barrier.wait()
    measure([("resnet34", 16, 512)], reps, f"[proc {rank}, cores {mine}] ", precision)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--procs", type=int, default=1)
    ap.add_argument("--cores-per-proc", type=int, default=2)
    ap.add_argument("--reps", type=int, default=9)
    ap.add_argument("--precision", default="f16x3_all")
    a = ap.parse_args()
    if a.procs <= 1:
        measure([("resnet34", 16, 512), ("efficientnet-b4", 4, 1024)], a.reps, "", a.precision)
    else:
        import multiprocessing as mp
        assert a.procs <= 6, "the GPU box admits at most 6 processes on its card"
        ctx = mp.get_context("spawn")
        bar = ctx.Barrier(a.procs)
        ps = [ctx.Process(target=worker, args=(r, a.cores_per_proc, a.reps, a.precision, bar)) for r in range(a.procs)]
        for p in ps: p.start()
        for p in ps: p.join()
        sys.exit(max(p.exitcode or 0 for p in ps))
