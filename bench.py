#!/usr/bin/env python3
"""Headline benchmark (BASELINE.json): train images/s of Unet(resnet34) at 512x512, batch 16 per GPU,
one process per GPU (torch.distributed over RCCL), synthetic data resident in HBM.

A "step" = zero_grad + forward + Dice loss + backward + (bucketed gradient all-reduce) + Adam — the
hot lines of /root/reference/src/train.py:86-105 — on configs[1] of BASELINE.json.

    python bench.py                       (= --gpus 1 --steps 100 --warmup 20)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \\
        --master-port P bench.py --gpus N --steps K --warmup W

Rank 0 prints ONE JSON line.  `roofline` is measured live with HIP event pairs attached to every conv / wgrad
dispatch (hipExtLaunchKernelGGL, on the stream the kernel is launched on) during the timed region (uwm_prof_*);
`roofline.frac` is ALGORITHMIC conv FLOP/s of the dominant kernel class over the dense peak of the matrix instruction it
issues (`roofline.mfma_util` = the FLOPs the pipe executed, 3x for the fp16x3 split products); `cpu_baseline` times
the CPU oracle (the reference's CPU path restated in plain torch) on this host's cores on a bounded
sample of the same workload.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

# The library overlaps its weight-gradient kernels on a second HIP stream.  ROCm multiplexes streams onto
# GPU_MAX_HW_QUEUES hardware queues (default 4) in creation order; once torch.distributed's RCCL group has
# created its streams, the library's side stream lands on the compute stream's queue and the overlap is lost
# (measured: 632 vs 668 img/s on the data-parallel path, profiles/r01 notes in DESIGN.md 6).  Must be set before
# the HIP runtime initialises.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

WINO_RATIO = 2.25            # direct-conv multiplies per Winograd F(2x2,3x3) multiply (36 / 16)
F32_MATRIX_PEAK_TFLOPS = 157.3      # MI355X_MICROARCH.md: v_mfma_f32_* peak (= fp32 vector peak)
BF16_MATRIX_PEAK_TFLOPS = 2500.0    # MI355X_MICROARCH.md: dense bf16 / f16 MFMA peak (v_mfma_f32_16x16x32_{f16,bf16})
F16X3_DTYPE = ("f32 storage and accumulation; the convolution products of the encoder and decoder layers (3x3, 7x7 stem, stride-2 and 1x1 "
               "layers: forward, data gradients, and the weight gradients of the stride-1 3x3 layers and the stem) as fp16x3 "
               "splits - every fp32 operand = two fp16 halves (22 mantissa bits), a*b = ah*bh + ah*bl + al*bh on "
               "v_mfma_f32_16x16x32_f16, exact power-of-two range scaling; the segmentation head, the stride-2 / 1x1 weight gradients "
               "and every other kernel exact fp32")


def cpu_baseline(encoder: str, hw: int, budget_s: float = 25.0, arch: str = "Unet", batch: int = 16, decoder_channels=None,
                 precision: str = "f32", routing_ref=None):
    """Oracle (torch CPU fp32) train step on this host's cores.  `value` is timed at the bench's OWN batch size (SURVEY 8d:
    the identical config-2 step, >= 3 timed steps) when one such step fits the time budget, else on a bs2 sample; the
    parity numbers (mask IoU, logits, loss, gradient cosine) always come from a bs2 sample of the same workload."""
    import torch
    from oracle import unet_oracle as O
    cores = len(os.sched_getaffinity(0))
    try:                                     # cgroup CPU quota (the GPU box shows 256 cores but grants 16)
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            cores = max(1, min(cores, int(int(q) / int(per))))
    except Exception:
        pass
    torch.set_num_threads(cores)
    cores = torch.get_num_threads()
    kw = {"decoder_channels": tuple(decoder_channels)} if decoder_channels else {}
    crit = O.DiceLoss(smooth=1e-5)

    def timed(bs_, min_steps, max_steps, budget):
        model = O.build(encoder, seed=42, arch=arch, **kw)
        model.train()
        opt = torch.optim.Adam(model.parameters(), lr=1e-4, weight_decay=1e-4)
        xs, ts = O.synthetic_batch(bs_, hw, hw, seed=42)
        tw = time.perf_counter()
        O.train_step(model, crit, opt, xs, ts)                      # warm-up
        tw = time.perf_counter() - tw
        n_, t0_ = 0, time.perf_counter()
        while n_ < min_steps or (time.perf_counter() - t0_ < budget and n_ < max_steps):
            O.train_step(model, crit, opt, xs, ts)
            n_ += 1
        return n_, time.perf_counter() - t0_, tw

    bs = 2
    n, dt, tw = timed(bs, 3, 8, budget_s / 2)
    out = {"value": round(bs * n / dt, 4), "unit": "images/s", "cores": cores, "kind": "port",
           "sample": f"{arch}-{encoder} {hw}x{hw} bs{bs} fwd+Dice+bwd+Adam on torch-CPU fp32 oracle, 1 warm-up + {n} timed steps"}
    # the bench's own batch size: 1 warm-up + 3 timed steps, only when the bs2 rate says that fits ~budget_s
    est = 4.0 * batch / max(out["value"], 1e-9)
    if batch > bs and est <= 1.6 * budget_s:
        try:
            nb, dtb, _ = timed(batch, 3, 3, 0.0)
            out["bs2_sample"] = {"value": out["value"], "sample": out["sample"]}
            out["value"] = round(batch * nb / dtb, 4)
            out["sample"] = (f"{arch}-{encoder} {hw}x{hw} bs{batch} (the benched step itself) fwd+Dice+bwd+Adam on torch-CPU fp32 oracle, "
                             f"1 warm-up + {nb} timed steps; parity numbers from a bs2 sample")
        except Exception as e:              # e.g. host memory: keep the bs2 figure and say so
            out["full_batch_note"] = f"bs{batch} oracle step not timed: {type(e).__name__}: {e}"
    elif batch > bs:
        out["full_batch_note"] = f"bs{batch} oracle steps would take ~{est:.0f} s on {cores} cores: bs2 sample only"
    x, t = O.synthetic_batch(bs, hw, hw, seed=42)
    # BASELINE.json's metric also names "mask IoU vs CPU ref": masks (logit > 0) of the sample batch from IDENTICAL
    # weights on both paths (train-mode BatchNorm).  Not compared after optimizer steps: Adam (eps 1e-8) turns the
    # zero-mean gradient noise of BatchNorm-invariant weight directions into +-lr steps, so any two fp32
    # implementations — two runs of torch itself with different thread counts included — walk apart.
    try:
        import unet_watermark_amd as U
        dev = torch.device("cuda", torch.cuda.current_device())
        ref0 = O.build(encoder, seed=42, arch=arch, **kw)
        hm = getattr(U, arch)(encoder, **kw).to(dev)
        # the parity numbers are those of the benched precision mode ON THE KERNELS THE BENCHED BATCH TAKES: the sample is 2 images,
        # the routing batch (uwm_set_routing_batch) makes every size-dependent kernel choice as for `batch` images
        hm.set_precision(precision, routing_batch=batch)
        hm.routing(enable=True)
        hm.load_state_dict(ref0.state_dict())
        hm.train(); ref0.train()
        cr, ch = O.DiceLoss(smooth=1e-5), U.DiceLoss(mode="binary", smooth=1e-5)
        o_ref = ref0(x); l_ref = cr(o_ref, t.unsqueeze(1)); l_ref.backward()
        o_hip = hm(x.to(dev)); l_hip = ch(o_hip, t.unsqueeze(1).to(dev)); l_hip.backward()
        lg, lr_ = o_hip.detach().cpu(), o_ref.detach()
        a, b = lg > 0, lr_ > 0
        inter, union = float((a & b).sum()), float((a | b).sum())
        out["mask_iou_vs_cpu_ref"] = round(inter / union, 6) if union else 1.0
        out["logit_max_abs_err_vs_cpu_ref"] = float(f"{float((lg - lr_).abs().max()):.3e}")
        out["loss_abs_diff_vs_cpu_ref"] = float(f"{abs(float(l_hip.detach()) - float(l_ref.detach())):.3e}")
        gref = dict(ref0.named_parameters()); cmin = 1.0
        for name, p_ in hm.named_parameters():
            g1, g2 = p_.grad.detach().cpu().double().flatten(), gref[name].grad.double().flatten()
            if float(g2.norm()) > 0 and float(g1.norm()) > 0:
                cmin = min(cmin, float(g1 @ g2 / (g1.norm() * g2.norm())))
        out["min_grad_cosine_vs_cpu_ref"] = round(cmin, 6)
        out["parity_precision_mode"] = precision
        rt = hm.routing()
        kinds = {}
        for pas, _layer, kern in rt:
            kinds[f"{pas}:{kern}"] = kinds.get(f"{pas}:{kern}", 0) + 1
        out["parity_sample_routing"] = {"routing_batch": batch, "launches": len(rt), "kernels": dict(sorted(kinds.items())),
                                        "matches_timed_step": (routing_ref is None or [tuple(r) for r in rt] == [tuple(r) for r in routing_ref])}
        hm.routing(enable=False)
        # the other precision modes on the same batch / weights / oracle run
        alt = {}
        for mode in [m_ for m_ in ("f32", "f16x3_all", "bf16x3", "f16x3_bwd2", "f16x1") if m_ != precision]:
            hm.load_state_dict(ref0.state_dict()); hm.set_precision(mode, routing_batch=batch)
            for p_ in hm.parameters():
                p_.grad = None
            o2 = hm(x.to(dev)); l2 = ch(o2, t.unsqueeze(1).to(dev)); l2.backward()
            c2 = 1.0
            for name, p_ in hm.named_parameters():
                g1, g2 = p_.grad.detach().cpu().double().flatten(), gref[name].grad.double().flatten()
                if float(g2.norm()) > 0 and float(g1.norm()) > 0:
                    c2 = min(c2, float(g1 @ g2 / (g1.norm() * g2.norm())))
            a2 = o2.detach().cpu() > 0
            alt[mode] = {"logit_max_abs_err_vs_cpu_ref": float(f"{float((o2.detach().cpu() - lr_).abs().max()):.3e}"),
                         "mask_iou_vs_cpu_ref": round(float((a2 & b).sum()) / max(1.0, float((a2 | b).sum())), 6),
                         "loss_abs_diff_vs_cpu_ref": float(f"{abs(float(l2.detach()) - float(l_ref.detach())):.3e}"),
                         "min_grad_cosine_vs_cpu_ref": round(c2, 6)}
        hm.set_precision(precision, routing_batch=0)
        out["alt_modes_parity"] = alt
    except Exception as e:                      # never let the checker break the bench line
        out["mask_iou_vs_cpu_ref"] = None
        out["mask_note"] = f"not computed: {type(e).__name__}: {e}"
    return out


def source_sha256() -> str:
    """sha256 over the kernel sources (csrc/*.hip, *.h, include/uwm.h, sorted): ties a committed PMC traffic file to the
    code that produced it (the GPU box has no .git, so a commit SHA cannot be recomputed there)."""
    import glob
    import hashlib
    h = hashlib.sha256()
    files = sorted(glob.glob(os.path.join(ROOT, "unet-watermark_amd", "csrc", "*.hip")) +
                   glob.glob(os.path.join(ROOT, "unet-watermark_amd", "csrc", "*.h")) + glob.glob(os.path.join(ROOT, "unet-watermark_amd", "csrc", "*.inc")) +
                   [os.path.join(ROOT, "include", "uwm.h")])
    for f in files:
        h.update(os.path.basename(f).encode()); h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def collect_prof(L, steps):
    """-> ({kernel: entry}, ncls) from uwm_prof_collect ({launches, ms, algorithmic FLOPs, algorithmic bytes} per class)."""
    prof = (C.c_double * (64 * 4))()
    ncls = L.lib().uwm_prof_collect(prof, 64)
    ents = {}
    for c in range(ncls):
        cnt, ms, fl, by = prof[c * 4], prof[c * 4 + 1], prof[c * 4 + 2], prof[c * 4 + 3]
        if cnt > 0 and ms > 0:
            name = L.lib().uwm_prof_class_name(c).decode()
            wino = "wino" in name or "up2" in name       # Winograd F(2x2,3x3) and the sub-pixel form of conv-after-upsample: the MFMA pipe executes 16 multiplies per 36 direct ones
            f16 = "f16x3" in name                        # direct form, three half-precision MFMAs per product block: 3x the direct FLOPs on the f16 pipe
            factor = (3.0 if f16 else 1.0) * (1.0 / WINO_RATIO if wino else 1.0)      # (the sub-pixel fp16x3 kernels: both)
            ents[name] = {"kernel": name, "launches_per_step": cnt / max(1, steps), "avg_us": round(1e3 * ms / cnt, 2),
                          "ms_per_step": round(ms / max(1, steps), 3),
                          "mfma_tflops": round(fl * factor / ms / 1e9, 2),
                          "mfma_pipe": "f16 (v_mfma_f32_16x16x32_f16)" if f16 else "f32 (v_mfma_f32_16x16x4_f32)",
                          "mfma_peak_tflops": BF16_MATRIX_PEAK_TFLOPS if f16 else F32_MATRIX_PEAK_TFLOPS,
                          "algorithmic_tflops": round(fl / ms / 1e9, 2),
                          "algorithmic_bytes_per_launch": int(by / cnt), "algorithmic_GBps": round(by / ms / 1e6, 1),
                          "_ms": ms, "_fl": fl, "_exec": fl * factor, "_peak": BF16_MATRIX_PEAK_TFLOPS if f16 else F32_MATRIX_PEAK_TFLOPS}
    return ents


def ddp_verdict(local, real_step, stages, world, rank, dev):
    """The collective half of validate_ddp, free of the model so that tests/test_ddp_gloo.py can drive it over gloo with fabricated
    arenas at world 8: `local` = this rank's local gradient arena; `real_step()` runs the step under test and returns (exchanged
    gradient arena, parameter arena).  Checks, bucket by bucket ([begin, end) ranges `stages`): the exchanged gradients equal ONE plain
    SUM all-reduce of the local ones within fp32 summation-order noise and are bit-identical across ranks (int64 checksums of the
    raw bits, all-gathered); the parameters afterwards are bit-identical across ranks; the ranks' local gradients differ when
    world > 1.  A failure on ANY rank prints the evidence and exits 3 on EVERY rank (the verdict is all-reduced first, so no rank is
    left waiting in a collective).  Returns (rows, local |g|_1 per rank, parameters identical)."""
    import torch
    import torch.distributed as dist
    expect = local.clone()
    dist.all_reduce(expect, op=dist.ReduceOp.SUM)
    lsum = torch.tensor([float(local.double().abs().sum())], dtype=torch.float64, device=dev)
    lall = [torch.zeros_like(lsum) for _ in range(world)]
    dist.all_gather(lall, lsum)
    lvals = [float(v) for v in lall]
    got, params = real_step()
    bits = got.view(torch.int32).to(torch.int64)
    rows, problems = [], []
    for k, (b, e) in enumerate(stages):
        if e <= b:
            continue
        ck = torch.stack([bits[b:e].sum(), (bits[b:e] * (torch.arange(e - b, device=dev) % 251 + 1)).sum()])
        allck = [torch.zeros_like(ck) for _ in range(world)]
        dist.all_gather(allck, ck)
        same = all(torch.equal(allck[0], c_) for c_ in allck)
        scale = float(expect[b:e].abs().max())
        err = float((got[b:e] - expect[b:e]).abs().max())
        ok_val = err <= 2e-5 * max(scale, 1e-30) + 1e-12
        rows.append({"bucket": k, "floats": e - b, "bit_identical_across_ranks": bool(same),
                     "max_abs_diff_vs_single_allreduce": float(f"{err:.3e}"), "max_abs": float(f"{scale:.3e}")})
        if not same:
            problems.append(f"bucket {k}: gradient bits differ across ranks after the all-reduce")
        if not ok_val:
            problems.append(f"bucket {k}: bucketed all-reduce differs from the single all-reduce of the local gradients by {err:.3e} (max {scale:.3e})")
        if not scale > 0:
            problems.append(f"bucket {k}: all-zero gradient")
    if world > 1 and len(set(lvals)) == 1:
        problems.append(f"local gradients are identical on every rank (|g|_1 = {lvals[0]}): the ranks are not training on rank-distinct data")
    # parameters after the first optimizer step
    pbits = params.view(torch.int32).to(torch.int64)
    pck = torch.stack([pbits.sum(), (pbits * (torch.arange(pbits.numel(), device=dev) % 251 + 1)).sum()])
    pall = [torch.zeros_like(pck) for _ in range(world)]
    dist.all_gather(pall, pck)
    psame = all(torch.equal(pall[0], c_) for c_ in pall)
    if not psame:
        problems.append("parameters differ across ranks after the first optimizer step")
    bad = torch.tensor([1.0 if problems else 0.0], device=dev)
    dist.all_reduce(bad, op=dist.ReduceOp.MAX)
    if float(bad) > 0:
        if problems:
            print(json.dumps({"ddp_check": "FAILED", "rank": rank, "problems": problems, "buckets": rows}), file=sys.stderr, flush=True)
        if dev.type == "cuda":
            dist.barrier(device_ids=[dev.index])
        else:
            dist.barrier()
        sys.exit(3)
    return rows, lvals, psame


def validate_ddp(trainer, model, x, t, world, rank, dev):
    """Self-validation of the data-parallel exchange, run once before the warm-up (nobody can watch an 8-GPU run):
    (1) every rank's local gradient arena (one backward, no collective) is SUM-all-reduced in ONE plain collective = the
        expected exchanged gradient; the ranks' local gradients must DIFFER (rank-distinct data) when world > 1;
    (2) one real Trainer.step (five bucketed all-reduces on the communication stream, overlapped with the staged backward)
        must leave, bucket by bucket, the same gradients: equal to (1) within fp32 summation-order noise, and BIT-identical
        across ranks (int64 checksum of the raw bits, all-gathered);
    (3) after that first optimizer step the parameter and BatchNorm-free state must be bit-identical across ranks.
    Any failure prints the evidence on rank 0 and exits non-zero on EVERY rank (the verdict itself is all-reduced, so no
    rank is left waiting in a collective)."""
    import torch
    import torch.distributed as dist
    from unet_watermark_amd import _lib as L
    m = model
    n, _, h, w = x.shape
    nst = len(m.stages)
    # the check leaves no trace: parameters, BatchNorm buffers and counters are put back afterwards, the optimizer restarts
    snap = (m.flat_parameters().clone(), m._buffer_arena.clone(), m._nbt_arena.clone())
    # (1) local backward of the same batch, no exchange, no optimizer step
    logits = m._forward_raw(x, training=True)
    trainer._buffers(n, h, w, dev)
    tc = t.contiguous()
    L.check(L.lib().uwm_loss(C.c_void_p(logits.data_ptr()), m._cp, C.c_void_p(tc.data_ptr()), L.target_dtype_code(tc), n * h * w,
                             trainer.w_dice, trainer.w_bce, trainer.smooth, trainer.eps, C.c_void_p(trainer._scratch.data_ptr()),
                             C.c_void_p(trainer._loss.data_ptr()), C.c_void_p(trainer._dl.data_ptr()), m._cp, 1.0,
                             C.c_void_p(L.stream_ptr(dev))))
    m._backward_raw(trainer._dl, 0, nst)
    torch.cuda.synchronize(dev)
    local = m.flat_grads().clone()
    # running statistics moved in (1): put every rank back on rank 0's buffers so (2) starts from replicas, as a run does
    dist.broadcast(m._buffer_arena, src=0)
    # (2) the real step
    def real_step():
        trainer.step(x, t)
        torch.cuda.synchronize(dev)
        return m.flat_grads(), m.flat_parameters()
    rows, lvals, psame = ddp_verdict(local, real_step, m.stages, world, rank, dev)
    m.flat_parameters().copy_(snap[0]); m._buffer_arena.copy_(snap[1]); m._nbt_arena.copy_(snap[2])
    for b_ in (trainer.opt._bufs or []):
        b_.zero_()
    trainer.opt._step = 0
    torch.cuda.synchronize(dev)
    return {"ranks": world, "buckets": rows, "params_bit_identical_after_step1": bool(psame),
            "local_grad_l1_per_rank": [float(f"{v:.6e}") for v in lvals]}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100, help="timed steps (SURVEY 8d protocol: 20 warm-up + 100 timed)")
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--batch", type=int, default=16, help="images per GPU")
    ap.add_argument("--size", type=int, default=512)
    ap.add_argument("--encoder", default="resnet34")
    ap.add_argument("--arch", default="Unet", choices=["Unet", "UnetPlusPlus"],
                    help="Unet = BASELINE.json's configs (headline); UnetPlusPlus = the reference's default MODEL.NAME (SURVEY 8 f3)")
    ap.add_argument("--decoder-channels", default=None,
                    help="comma-separated MODEL.DECODER_CHANNELS (default smp's 256,128,64,32,16); the reference's large YAML "
                         "(/root/reference/src/configs/unet_watermark_large.yaml:5-19,36) is --arch UnetPlusPlus --encoder resnet50 "
                         "--decoder-channels 1024,512,256,128,64 --size 1024 --batch 8")
    ap.add_argument("--no-ddp-check", action="store_true", help="skip the warm-up self-validation of the data-parallel exchange")
    ap.add_argument("--precision", default="f16x3_all", choices=["f32", "f16x3", "f16x3_all", "bf16x3", "bf16x3_all", "f16x3_bwd2", "f16x1"],
                    help="precision mode of the timed steps (uwm_set_precision).  f16x3_all (default): fp16x3 split products on the "
                         "3x3 stride-1 convolutions, fp32-class accuracy - it meets the fp32 mode's parity bars (tests); f32: every "
                         "product on the exact-fp32 matrix instruction (reported under alt_modes otherwise)")
    ap.add_argument("--graph", action="store_true",
                    help="replay the train step from ONE captured hipGraph (Trainer(use_graph=True); single process only)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-overlap", action="store_true", help="all-reduce after the whole backward")
    ap.add_argument("--prof-steps", type=int, default=5, help="timed steps (the last ones) whose conv launches carry HIP event pairs")
    ap.add_argument("--alt-steps", type=int, default=20, help="timed steps per opt-in precision mode (alt_modes; 0 = skip; N=1 only)")
    ap.add_argument("--serial-steps", type=int, default=3,
                    help="extra UNTIMED steps after the timed region with the wgrad side stream off, to report the "
                         "dominant kernel alone next to its in-step (co-resident) figure (0 = skip)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    import unet_watermark_amd as U
    from unet_watermark_amd import _lib as L
    from unet_watermark_amd.train import Trainer

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        # fail loudly: a line that says n_gpus=N must come from N ranks (one per GPU) launched by torch.distributed.run
        raise SystemExit(f"bench.py --gpus {args.gpus} but WORLD_SIZE={world}: launch N>1 as `python -m torch.distributed.run --nnodes=1 "
                         f"--nproc-per-node {args.gpus} --master-addr 127.0.0.1 --master-port P bench.py --gpus {args.gpus} ...`")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (the hot path has no CPU fallback)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    force_ddp = bool(int(os.environ.get("UWM_FORCE_DDP", "0")))      # 1-rank RCCL group: exercises the DDP path on one GPU
    if world > 1 or force_ddp:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
        dist.init_process_group(backend="nccl", **({} if os.environ.get("UWM_PG_LAZY") else {"device_id": dev}))
        if dist.get_world_size() != args.gpus or dist.get_backend() != "nccl":
            raise SystemExit(f"bench.py --gpus {args.gpus}: process group has {dist.get_world_size()} ranks on backend {dist.get_backend()!r}")
        if world > 1 and torch.cuda.device_count() < world // max(1, int(os.environ.get("NNODES", "1"))):
            raise SystemExit(f"bench.py --gpus {world}: only {torch.cuda.device_count()} HIP devices visible (one rank per GPU)")

    torch.manual_seed(42)                                  # identical init on every rank (+ broadcast in Trainer)
    dec = tuple(int(c) for c in args.decoder_channels.split(",")) if args.decoder_channels else None
    model = getattr(U, args.arch)(args.encoder, encoder_weights=None, in_channels=3, classes=1,
                                  **({"decoder_channels": dec} if dec else {})).to(dev)
    model.set_precision(args.precision)
    trainer = Trainer(model, w_dice=1.0, w_bce=0.0, smooth=1e-5, lr=1e-4, weight_decay=1e-4,
                      overlap_comm=not args.no_overlap, force_ddp=force_ddp and not os.environ.get("UWM_PG_ONLY"),
                      use_graph=args.graph and world == 1 and not force_ddp)
    g = torch.Generator(device="cpu").manual_seed(42 + rank)      # rank-distinct synthetic data
    n, s = args.batch, args.size
    x = torch.randn(n, 3, s, s, generator=g).to(dev)
    t = torch.zeros(n, s, s, dtype=torch.int64)
    for i in range(n):                                             # seeded rectangles, 5-20 % area
        frac = 0.05 + 0.15 * float(torch.rand((), generator=g))
        rh = max(1, int((frac * s * s) ** 0.5)); rw = max(1, min(s, int(frac * s * s / rh)))
        y0 = int(torch.randint(0, s - rh + 1, (), generator=g)); x0 = int(torch.randint(0, s - rw + 1, (), generator=g))
        t[i, y0:y0 + rh, x0:x0 + rw] = 1
    t = t.to(dev)

    def barrier():
        if world > 1:
            dist.barrier(device_ids=[local_rank])
        torch.cuda.synchronize(dev)

    ddp_check = None
    if (world > 1 or force_ddp) and trainer.ddp and not args.no_ddp_check:
        ddp_check = validate_ddp(trainer, model, x, t, world, rank, dev)        # exits non-zero on every rank if it fails
    for _ in range(args.warmup):
        loss = trainer.step(x, t)
    barrier()
    # profiled launches go through hipExtLaunchKernelGGL with an event pair each (a little host work per launch), so only
    # the LAST `prof_steps` steps of the timed region carry them; `kernels` / `roofline` come from those
    prof_steps = min(args.prof_steps, args.steps)
    t0 = time.perf_counter()
    for i in range(args.steps):
        if i == args.steps - prof_steps:
            L.lib().uwm_prof_enable(1)
            trainer.use_graph = False              # (profiled launches carry dispatch-attached events: eager steps)
        loss = trainer.step(x, t)
    barrier()
    dt = time.perf_counter() - t0
    L.lib().uwm_prof_enable(0)
    ents = collect_prof(L, prof_steps)
    loss_val = float(loss[0].item())
    # kernel routing of the benched step (one extra UNTIMED step with the library's routing record on): the parity sample of
    # cpu_baseline must take the same kernels layer by layer
    routing_ref = None
    if rank == 0 and world == 1:
        model.routing(enable=True)
        ug = trainer.use_graph
        trainer.use_graph = False
        trainer.step(x, t); torch.cuda.synchronize(dev)
        trainer.use_graph = ug
        routing_ref = model.routing()
        model.routing(enable=False)
    # the dominant kernel ALONE (not part of `value`): the same step with the wgrad side stream switched off
    ents_s = None
    if args.serial_steps > 0 and rank == 0 and world == 1:
        L.lib().uwm_set_side_stream(model._h, 0)
        trainer.step(x, t); torch.cuda.synchronize(dev)
        L.lib().uwm_prof_enable(1)
        for _ in range(args.serial_steps):
            trainer.step(x, t)
        torch.cuda.synchronize(dev)
        L.lib().uwm_prof_enable(0)
        ents_s = collect_prof(L, args.serial_steps)
        L.lib().uwm_set_side_stream(model._h, 1)

    # the other precision modes (NOT part of `value`): the same step with uwm_set_precision(...)
    alt_modes = None
    if args.alt_steps > 0 and rank == 0 and world == 1:
        alt_modes = {}
        for mode in [m_ for m_ in ("f32", "f16x3_all", "bf16x3", "f16x3_bwd2", "f16x1") if m_ != args.precision]:
            model.set_precision(mode)
            for _ in range(3):
                trainer.step(x, t)
            torch.cuda.synchronize(dev)
            ta = time.perf_counter()
            for _ in range(args.alt_steps):
                la = trainer.step(x, t)
            torch.cuda.synchronize(dev)
            dta = time.perf_counter() - ta
            alt_modes[mode] = {"value": round(n * args.alt_steps / dta, 2), "unit": "images/s", "ms_per_step": round(1e3 * dta / args.alt_steps, 3),
                               "steps": args.alt_steps, "loss": round(float(la[0].item()), 6),
                               "dtype": {"f32": "f32: every product on the exact-fp32 matrix instruction (v_mfma_f32_16x16x4_f32)",
                                         "f16x3_all": F16X3_DTYPE,
                                         "bf16x3": "bf16x3 split products (16-bit operands) on the dgrads of the 3x3 stride-1 layers; the rest f32",
                                         "f16x3_bwd2": "REDUCED precision in the backward only: forward = f16x3_all (logits identical, inside the 1e-3 bar); "
                                                       "dgrad / wgrad take dY as ONE fp16 (two split products per tile)",
                                         "f16x1": "REDUCED precision: one fp16 product per tile (hi*hi'), fp32 accumulation / storage / BatchNorm / optimizer - the "
                                                  "reference's own GPU arithmetic (fp16 autocast, /root/reference/src/train.py:75,89-98); logits OUTSIDE the "
                                                  "1e-3 bar (parity_vs_cpu_ref), never the headline"}[mode],
                               "reduced_precision": mode in ("f16x1", "f16x3_bwd2", "bf16x3")}
        model.set_precision(args.precision)

    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())

    if rank == 0:
        fwd, fwdbwd = model.conv_flops(s, s)
        kernels = sorted(ents.values(), key=lambda k: -k["ms_per_step"])
        tot_ms = sum(k["_ms"] for k in kernels); tot_fl = sum(k["_fl"] for k in kernels); tot_ex = sum(k["_exec"] for k in kernels)
        dom = kernels[0] if kernels else None
        ms_step = 1e3 * dt / args.steps
        headline = args.arch == "Unet" and args.encoder == "resnet34" and s == 512 and n == 16
        out = {
            "metric": "train_images_per_sec", "value": round(world * n * args.steps / dt, 2), "unit": "images/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_step, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": F16X3_DTYPE if args.precision == "f16x3_all" else args.precision, "precision_mode": args.precision, "data": "synthetic",
            "config": {"workload": f"{args.arch}-{args.encoder} {s}x{s} bs{n}/GPU train step: fwd + Dice + bwd + Adam "
                                   + (f"(BASELINE.json configs[{1 if world == 1 else 2}])" if headline else
                                      "(the per-GPU workload of BASELINE.json configs[3]: SURVEY 8 a18)"
                                      if (args.arch == "Unet" and args.encoder == "efficientnet-b4" and s == 1024 and n == 4) else
                                      "(not a BASELINE config: SURVEY 8 f3 / a18 widening)"),
                       "global_batch": world * n, "image": [s, s], "parallelism": f"dp{world}",
                       "decoder_channels": list(model.decoder_channels),
                       "grad_allreduce": ("rccl, 5 buckets overlapped with backward" if (world > 1 or force_ddp) else "none")},
            "loss": round(loss_val, 6),
            "hipgraph_step": bool(trainer.use_graph and not trainer.ddp),
            "rccl_ranks": (dist.get_world_size() if (world > 1 or force_ddp) else 0),
            "ddp_check": ddp_check,
            "model_tflops": round(world * n * fwdbwd * args.steps / dt / 1e12, 2),
            "roofline": None,
        }
        if dom:
            P = dom["_peak"]               # the dominant kernel's own matrix pipe (fp32 MFMA 157.3, f16 MFMA 2500)
            alone = None
            if ents_s is not None and dom["kernel"] in ents_s:
                a_ = ents_s[dom["kernel"]]
                alone = {"avg_launch_us": a_["avg_us"], "achieved": a_["algorithmic_tflops"], "frac": round(a_["algorithmic_tflops"] / P, 4),
                         "mfma_util": round(a_["mfma_tflops"] / P, 4), "mfma_executed_tflops": a_["mfma_tflops"],
                         "note": f"{args.serial_steps} extra untimed steps with the wgrad side stream off (nothing co-resident)"}
            traffic = None
            try:   # HBM bytes per launch from SEPARATE rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command
                   # (FETCH_SIZE doubled: MI355X_MICROARCH.md's gfx950 correction), digested by scripts/pmc_summary.py.
                   # Only a file made from THESE kernel sources is quoted.
                pm = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
                sha = source_sha256()
                if pm.get("source_sha256") != sha:
                    traffic = {"hbm_bytes_per_launch": None,
                               "note": f"profiles/pmc_traffic.json was measured on kernel sources {pm.get('source_sha256')} "
                                       f"(git {pm.get('git_sha')}), this run is {sha}: not quoted"}
                else:
                    # a profiler class may lump template instances / variants that rocprof names separately (conv_f16x3_kernel<4>, <2>,
                    # <1>, conv_f16x3s_kernel): launch-weighted average over them
                    k_ = dom["kernel"]
                    es_ = [v for n_, v in pm["kernels"].items() if n_ == k_ or n_.startswith(k_ + "<") or (k_ == "conv_f16x3_kernel" and n_ == "conv_f16x3s_kernel")]
                    e_ = None
                    if es_:
                        wsum = sum(max(1, v.get("launches", 1)) for v in es_)
                        e_ = {f_: int(sum(v[f_] * max(1, v.get("launches", 1)) for v in es_) / wsum)
                              for f_ in ("hbm_bytes_per_launch", "fetch_bytes_per_launch", "write_bytes_per_launch")}
                    if e_:
                        hb = e_["hbm_bytes_per_launch"]
                        traffic = {"hbm_bytes_per_launch": hb, "algorithmic_bytes_per_launch": dom["algorithmic_bytes_per_launch"],
                                   "ratio": round(hb / max(1, dom["algorithmic_bytes_per_launch"]), 3),
                                   "fetch_bytes_per_launch": e_.get("fetch_bytes_per_launch"), "write_bytes_per_launch": e_.get("write_bytes_per_launch"),
                                   "source_sha256": sha, "git_sha": pm.get("git_sha"), "source": pm["source"]}
            except Exception as e:
                traffic = {"hbm_bytes_per_launch": None, "note": f"no PMC traffic file: {type(e).__name__}"}
            out["roofline"] = {
                "bound": "mfma", "kernel": dom["kernel"], "achieved": dom["algorithmic_tflops"], "peak": P, "unit": "TFLOP/s",
                "frac": round(dom["algorithmic_tflops"] / P, 4),
                "mfma_util": round(dom["mfma_tflops"] / P, 4), "mfma_executed_tflops": dom["mfma_tflops"],
                "definition": "achieved = ALGORITHMIC FLOPs per launch (direct-convolution FLOPs of SURVEY.md 8(d): 2*N*Ho*Wo*Cout*Cin*k*k) "
                              "/ average launch duration of the kernel class with the largest share of kernel time, over the profiled "
                              "steps of the timed region (dispatch-attached HIP events on the launch stream; co-resident with the other "
                              "stream's kernels); peak = the dense peak of the matrix instruction THAT kernel issues "
                              "(MI355X_MICROARCH.md: v_mfma_f32_16x16x4_f32 157.3 TF, v_mfma_f32_16x16x32_f16 2500 TF); frac = achieved / "
                              "peak.  mfma_util = FLOPs the MFMA pipe EXECUTED / the same duration / peak: an fp16x3 kernel executes 3x its "
                              "algorithmic FLOPs (hi*hi + hi*lo + lo*hi buy fp32-class accuracy on the f16 pipe), a Winograd kernel "
                              "1/2.25 of them — pipe utilisation, not the roofline fraction",
                "mfma_pipe": dom["mfma_pipe"],
                "hbm_view": {"achieved_GBps": dom["algorithmic_GBps"], "peak_GBps": 8000.0, "frac": round(dom["algorithmic_GBps"] / 8000.0, 4),
                             "note": "algorithmic bytes (every operand once + the output once) / the same duration: the fp16x3 kernels are "
                                     "closer to this roof than to their matrix pipe's"},
                "avg_launch_us": dom["avg_us"], "launches_per_step": dom["launches_per_step"], "ms_per_step": dom["ms_per_step"],
                "alone": alone,
                # the next kernels by time in the step (the first three are within a few per cent of each other on the headline
                # config, so which one is "dominant" can flip between runs): same definitions
                "runners_up": [{"kernel": k["kernel"], "frac": round(k["algorithmic_tflops"] / k["_peak"], 4), "mfma_util": round(k["mfma_tflops"] / k["_peak"], 4),
                                "mfma_pipe": k["mfma_pipe"], "avg_launch_us": k["avg_us"],
                                "launches_per_step": k["launches_per_step"], "ms_per_step": k["ms_per_step"], "hbm_frac": round(k["algorithmic_GBps"] / 8000.0, 4),
                                "alone_frac": (round(ents_s[k["kernel"]]["algorithmic_tflops"] / k["_peak"], 4)
                                               if ents_s is not None and k["kernel"] in ents_s else None),
                                "alone_avg_launch_us": (ents_s[k["kernel"]]["avg_us"]
                                                        if ents_s is not None and k["kernel"] in ents_s else None)}
                               for k in kernels[1:3]],
                "step": {"algorithmic_tflops": round(tot_fl * 1e-9 / prof_steps / ms_step, 2),
                         "frac": round(sum(k["_fl"] / k["_peak"] for k in kernels) * 1e-9 / prof_steps / ms_step, 4),
                         "frac_definition": "sum over conv / wgrad classes of (algorithmic FLOPs / that class's matrix-pipe peak) / wall step time",
                         "mfma_executed_tflops": round(tot_ex * 1e-9 / prof_steps / ms_step, 2),
                         "mfma_util": round(sum(k["_exec"] / k["_peak"] for k in kernels) * 1e-9 / prof_steps / ms_step, 4),
                         "conv_kernel_ms_per_step": round(tot_ms / prof_steps, 3),
                         "note": "all conv / wgrad launches of a step: executed MFMA FLOPs / WALL step time (two streams overlap, so "
                                 "kernel ms per step may exceed the step)"},
                "traffic": traffic,
                "profiled_steps": f"last {prof_steps} of the {args.steps} timed steps"}
            out["kernels"] = [{k: v for k, v in e.items() if not k.startswith("_")} for e in kernels]
        if alt_modes is not None:
            out["alt_modes"] = alt_modes
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.encoder, s, arch=args.arch, batch=n, decoder_channels=dec, precision=args.precision,
                                               routing_ref=routing_ref)
            if alt_modes is not None:
                for mode, par in out["cpu_baseline"].pop("alt_modes_parity", {}).items():
                    alt_modes[mode]["parity_vs_cpu_ref"] = par
        print(json.dumps(out), flush=True)
    if world > 1 or force_ddp:
        dist.barrier(device_ids=[local_rank])
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
