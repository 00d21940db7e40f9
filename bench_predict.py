#!/usr/bin/env python3
"""BASELINE config 5: batch inference, 4096 synthetic 512x512 images in 64 batches of 64 on one GPU, eval-mode
forward (BatchNorm from running statistics) captured once in a hipGraph and replayed; masks by the
reference's raw-logit threshold.  Prints one JSON line.  (The driver's headline bench is bench.py.)"""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=64); ap.add_argument("--batches", type=int, default=64)
    ap.add_argument("--size", type=int, default=512); ap.add_argument("--encoder", default="resnet34")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--precision", default="f16x3", choices=["f32", "f16x3"],
                    help="f16x3 (default): fp16x3 split products on the 3x3 convolutions, fp32-class accuracy; f32: exact-fp32 matrix instruction")
    a = ap.parse_args()
    from unet_watermark_amd.predict import WatermarkPredictor
    from unet_watermark_amd.config import get_cfg_defaults
    cfg = get_cfg_defaults(); cfg.MODEL.NAME = "Unet"; cfg.MODEL.ENCODER_NAME = a.encoder      # BASELINE configs[4] names Unet
    torch.manual_seed(42)
    pred = WatermarkPredictor(config=cfg, device="cuda", precision=a.precision)
    x = torch.randn(a.batch, 3, a.size, a.size, device="cuda")
    for _ in range(2):
        m = pred.predict_mask(x, use_graph=not a.no_graph)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.batches):
        m = pred.predict_mask(x, use_graph=not a.no_graph)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    fwd, _ = pred.model.conv_flops(a.size, a.size)
    n = a.batch * a.batches
    # per-image equality of the batched, graph-replayed path with the batch-1 eager path
    lb = pred.logits(x, use_graph=not a.no_graph).clone()
    eq = True
    for i in (0, a.batch // 2, a.batch - 1):
        eq = eq and bool(torch.equal(pred.logits(x[i:i + 1], use_graph=False)[0], lb[i]))
    print(json.dumps({"metric": "predict_images_per_sec", "value": round(n / dt, 2), "unit": "images/s", "n_gpus": 1,
                      "images": n, "batch": a.batch, "ms_per_batch": round(1e3 * dt / a.batches, 3),
                      "dtype": "f32" if a.precision == "f32" else "f32 storage / accumulation, 3x3 conv products as fp16x3 splits (22-bit operands) on v_mfma_f32_16x16x32_f16",
                      "precision_mode": a.precision,
                      "data": "synthetic", "hipgraph": not a.no_graph,
                      "config": {"workload": f"Unet-{a.encoder} {a.size}x{a.size} eval forward + logit threshold, bs{a.batch} (BASELINE configs[4])"},
                      "model_tflops": round(n * fwd / dt / 1e12, 2), "mask_positive_frac": round(float((m > 0).float().mean()), 4),
                      "bitwise_equal_to_batch1_path": eq}))


if __name__ == "__main__":
    main()
