"""Gate for a NON-fused Winograd F(2x2,3x3) on the deep stages: input transform -> 16 batched GEMMs ->
inverse transform.  Times the GEMM phase on the persistent LDS-DMA GEMM (conv_gemm, via the 1x1 op entry point): the 16 GEMMs
of [tiles x Cin] x [Cin x Cout] are run as ONE 1x1 conv over 16 * tiles pixels (same FLOPs, same tile count; the real thing
would stream 16 filter slices instead of one), next to the fused kernels' time for the same layer (forward conv, 3x3 s1).
Transform passes are priced from their bytes (V and M: 4x the activation each, written once + read once)."""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tests.util import src, P, stream, rup
from unet_watermark_amd import _lib as L
dev = torch.device("cuda:0")
N = 16


def timeit(f):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 50      # us


for name, ch, h in (("layer2", 128, 64), ("layer3", 256, 32), ("layer4", 512, 16)):
    tiles = N * (h // 2) * (h // 2)
    # GEMM phase: M = 16 * tiles rows as an (N, hh, hh) image with hh*hh*N = 16*tiles
    hh = 2 * h                                   # N * (2h)^2 = 16 * N * (h/2)^2
    x = torch.randn(N, hh, hh, ch, device=dev); y = torch.empty(N, hh, hh, ch, device=dev)
    w = torch.randn(ch, rup(ch, 32), device=dev)
    s0 = src(x)
    g = lambda: L.check(L.lib().uwm_op_conv(C.byref(s0), None, P(w), ch, rup(ch, 32), 1, 1, 1, 0, N, ch, None, P(y), None, -1, stream()))
    us_g = timeit(g)
    fl_exec = 2.0 * 16 * tiles * ch * ch
    # fused kernel on the real layer (3x3 s1 p1, forward, with BatchNorm statistics)
    xi = torch.randn(N, h, h, ch, device=dev); yo = torch.empty(N, h, h, ch, device=dev)
    w3 = torch.randn(ch, rup(9 * ch, 32), device=dev); st = torch.zeros(2 * ch, dtype=torch.float64, device=dev)
    s1 = src(xi)
    f = lambda: L.check(L.lib().uwm_op_conv(C.byref(s1), None, P(w3), ch, rup(9 * ch, 32), 3, 3, 1, 1, N, ch, None, P(yo), P(st), -1, stream()))
    us_f = timeit(f)                            # (includes the op entry's filter-transform launch: a few us)
    act = N * h * h * ch * 4
    tr_bytes = 2 * (4 * act) * 2 + 2 * act       # V written + read, M written + read, activation read + output written
    print(f"{name}: {ch}->{ch} at {h}x{h} x{N}: GEMM phase {us_g:6.1f} us ({fl_exec / us_g / 1e6:5.1f} TF executed = {fl_exec / us_g / 1e6 / 157.3:.2f} of peak)"
          f" + transforms {tr_bytes / 1e6:5.0f} MB = {tr_bytes / 4.5e6:5.1f} us at 4.5 TB/s -> {us_g + tr_bytes / 4.5e6:6.1f} us in 3 launches"
          f" | fused kernel {us_f:6.1f} us")
