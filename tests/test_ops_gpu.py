"""GPU parity of libuwm's single-operator entry points against torch CPU fp32 primitives
(the oracle's primitives — SURVEY.md §8c).  Tolerances: fp32 accumulation-order noise only
(exact-fp32 MFMA), stated per test."""
import ctypes as C

import pytest
import torch
import torch.nn.functional as F

from tests.util import nhwc, nchw, pack_w, unpack_w, src, P, stream, rup

pytestmark = pytest.mark.gpu


def lib():
    from unet_watermark_amd import _lib as L
    return L


def _conv_case(dev, n, cin, cout, h, w, k, stride, pad, cfg=-1, lazy=False, seed=0):
    L = lib()
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(n, cin, h, w, generator=g)
    wt = torch.randn(cout, cin, k, k, generator=g) * (1.0 / (cin * k * k) ** 0.5)
    xin = x
    scale = shift = None
    if lazy:
        scale = torch.rand(rup(cin, 4), generator=g) + 0.5
        scale[::3] *= -1
        shift = torch.randn(rup(cin, 4), generator=g) * 0.3
        xin = torch.relu(x * scale[:cin, None, None] + shift[:cin, None, None])
    ref = F.conv2d(xin, wt, None, stride, pad)
    xd = nhwc(x).to(dev)
    wp, kpad = pack_w(wt)
    wp = wp.to(dev)
    coutp = rup(cout, 4)
    ho, wo = ref.shape[-2:]
    y = torch.full((n, ho, wo, coutp), float("nan"), device=dev)
    stats = torch.zeros(2 * coutp, dtype=torch.float64, device=dev)
    sc = scale.to(dev) if lazy else None
    sh = shift.to(dev) if lazy else None
    s0 = src(xd, sc, sh, relu=1 if lazy else 0)
    L.check(L.lib().uwm_op_conv(C.byref(s0), None, P(wp), cout, kpad, k, k, stride, pad, n, coutp, None, P(y), P(stats),
                                cfg, stream()))
    torch.cuda.synchronize()
    got = nchw(y.cpu(), cout)
    tol = 2e-5 * max(1.0, float(ref.abs().max()))
    assert torch.isfinite(y).all()
    assert (got - ref).abs().max() < tol, f"conv max err {(got - ref).abs().max()}"
    if coutp > cout:
        assert (y[..., cout:] == 0).all()
    # BatchNorm statistics of the output
    ssum = stats[:cout].cpu()
    ssq = stats[coutp:coutp + cout].cpu()
    rs = ref.double().sum((0, 2, 3))
    rq = (ref.double() ** 2).sum((0, 2, 3))
    assert torch.allclose(ssum, rs, rtol=1e-5, atol=1e-3 * max(1.0, float(rs.abs().max())) * 1e-2)
    assert torch.allclose(ssq, rq, rtol=1e-5, atol=1e-5 * float(rq.abs().max()))


@pytest.mark.parametrize("cfg", [0, 1, 2, 3, 4, 5])
def test_conv3x3_all_tile_configs(cuda, cfg):
    # layer1.0.conv1-like (reduced): every tile configuration must give the same answer,
    # including tiles that over-hang M and Cout.
    _conv_case(cuda, 2, 64, 64, 24, 40, 3, 1, 1, cfg=cfg)


@pytest.mark.parametrize("bn,cout", [(16, 16), (32, 32), (64, 64), (128, 128), (64, 96), (128, 256)])
def test_conv_patch_tile_configs(cuda, bn, cout):
    """patch-tiled 3x3 kernel (conv_patch.hip), every channel-tile config; 24x40 pixels -> partial
    8x16 tiles in W; Cout 96 over-hangs the 64-channel tile."""
    _conv_case(cuda, 2, 64, cout, 24, 40, 3, 1, 1, cfg=100 + bn)
    _conv_case(cuda, 1, 32, cout, 8, 16, 3, 1, 1, cfg=100 + bn, lazy=True, seed=2)


@pytest.mark.parametrize("cin,cout", [(64, 64), (64, 96), (32, 256), (128, 32), (16, 16), (16, 1), (8, 48)])
def test_conv_winograd(cuda, cin, cout):
    """Winograd F(2x2,3x3) kernel (conv_wino.hip), every channel-tile config (BN 64/32/16, over-hanging Cout);
    24x40 and 9x17 pixels -> partial 8x16 workgroup tiles and partial 2x2 Winograd tiles; lazy BatchNorm+ReLU
    input; BatchNorm statistics of the output."""
    _conv_case(cuda, 2, cin, cout, 24, 40, 3, 1, 1, cfg=300)
    _conv_case(cuda, 1, cin, cout, 9, 17, 3, 1, 1, cfg=300, lazy=True, seed=2)


@pytest.mark.parametrize("cin,cout", [(64, 64), (32, 160), (8, 64), (128, 96)])
def test_conv_winograd_8wave(cuda, cin, cout):
    """512-thread / 16x16-pixel Winograd variant (conv_wino8.hip): partial tiles in both directions, over-hanging
    Cout, single-chunk K (cin 8), lazy input, BatchNorm statistics."""
    _conv_case(cuda, 2, cin, cout, 24, 40, 3, 1, 1, cfg=308)
    _conv_case(cuda, 1, cin, cout, 17, 33, 3, 1, 1, cfg=308, lazy=True, seed=2)


@pytest.mark.parametrize("cin,cout,h,w", [(16, 32, 16, 32), (32, 40, 24, 48), (64, 64, 32, 32), (48, 16, 8, 16), (128, 96, 16, 16)])
def test_conv_winograd_bf16x3(cuda, cin, cout, h, w):
    """conv_wino_x3 (cfg 400): Winograd F(2x2,3x3) with every product as a 3-term split-bf16 MFMA sum (the opt-in bf16x3
    precision mode) against F.conv2d in fp32 — over-hanging Cout (40, 96 on 32-channel tiles), partial pixel tiles, bias,
    BatchNorm statistics.  Bar: 1e-4 of the output scale (measured ~1e-5: ~16 mantissa bits per operand)."""
    L = lib()
    g = torch.Generator().manual_seed(cin * 7 + cout)
    n = 2
    x = torch.randn(n, cin, h, w, generator=g)
    wt = torch.randn(cout, cin, 3, 3, generator=g) * (2.0 / (9 * cin)) ** 0.5
    bias = torch.randn(cout, generator=g)
    y = torch.nn.functional.conv2d(x, wt, bias, padding=1)
    xd = nhwc(x.to(cuda)); wp, kpad = pack_w(wt.to(cuda)); cp = rup(cout, 4)
    bp = torch.zeros(cp, device=cuda); bp[:cout] = bias.to(cuda)
    out = torch.full((n, h, w, cp), float("nan"), device=cuda)
    stats = torch.zeros(2 * cp, dtype=torch.float64, device=cuda)
    s0 = src(xd)
    L.check(L.lib().uwm_op_conv(C.byref(s0), None, P(wp), cout, kpad, 3, 3, 1, 1, n, cp, P(bp), P(out), P(stats), 400, stream()))
    got = nchw(out.cpu(), cout)
    err = float((got - y).abs().max()); scale = max(1.0, float(y.abs().max()))
    assert err < 1e-4 * scale, (err, scale)
    assert torch.isfinite(out[..., :cout]).all()
    ssum = stats[:cout].cpu(); ref_sum = y.double().sum((0, 2, 3))
    assert torch.allclose(ssum, ref_sum, rtol=1e-4, atol=1e-3 * float(ref_sum.abs().max() + 1))


def test_conv_winograd_bf16x3_lazy_upsample_concat(cuda):
    """the bf16x3 kernel behind the decoder's two-source loader: nearest x2 upsample of a lazily normalised + ReLU-ed tensor
    concatenated with a skip tensor (16-channel chunk boundary at the concat)."""
    L = lib()
    g = torch.Generator().manual_seed(5)
    n, c0, c1, cout, h, w = 2, 32, 16, 32, 16, 32
    a = torch.randn(n, c0, h // 2, w // 2, generator=g); sk = torch.randn(n, c1, h, w, generator=g)
    sc = torch.rand(c0, generator=g) + 0.5; sh = torch.randn(c0, generator=g) * 0.3
    wt = torch.randn(cout, c0 + c1, 3, 3, generator=g) * 0.05
    act = torch.relu(a * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1))
    xin = torch.cat([torch.nn.functional.interpolate(act, scale_factor=2, mode="nearest"), sk], 1)
    y = torch.nn.functional.conv2d(xin, wt, None, padding=1)
    ad, skd = nhwc(a.to(cuda)), nhwc(sk.to(cuda)); wp, kpad = pack_w(wt.to(cuda))
    scd, shd = sc.to(cuda), sh.to(cuda)
    out = torch.empty(n, h, w, cout, device=cuda)
    s0 = src(ad, scd, shd, relu=1, up=1); s1 = src(skd)
    L.check(L.lib().uwm_op_conv(C.byref(s0), C.byref(s1), P(wp), cout, kpad, 3, 3, 1, 1, n, cout, None, P(out), None, 400, stream()))
    err = float((nchw(out.cpu(), cout) - y).abs().max())
    assert err < 1e-4 * max(1.0, float(y.abs().max())), err


@pytest.mark.parametrize("n,hs,ws,lazy,cfg", [(2, 16, 32, True, 700), (1, 8, 16, False, 700), (3, 24, 48, True, -1)])
def test_conv_up2_subpixel_forward(cuda, n, hs, ws, lazy, cfg):
    """decoder block 4 conv1: 3x3 over a nearest-x2 upsampled (lazily normalised + ReLU-ed) 32-channel tensor, 16 outputs,
    as four 2x2 parity-class convolutions on the low-resolution tensor (conv_up2.hip); cfg -1 = the auto route must pick it.
    Output and BatchNorm statistics against torch's interpolate + conv2d."""
    L = lib()
    g = torch.Generator().manual_seed(23 + hs)
    a = torch.randn(n, 32, hs, ws, generator=g)
    sc = torch.rand(32, generator=g) + 0.5; sh = torch.randn(32, generator=g) * 0.3
    wt = torch.randn(16, 32, 3, 3, generator=g) * 0.08
    act = torch.relu(a * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1)) if lazy else a
    y = F.conv2d(F.interpolate(act.double(), scale_factor=2, mode="nearest"), wt.double(), None, padding=1)
    ad = nhwc(a.to(cuda)); wp, kpad = pack_w(wt.to(cuda))
    out = torch.full((n, 2 * hs, 2 * ws, 16), float("nan"), device=cuda)
    stats = torch.zeros(32, dtype=torch.float64, device=cuda)
    t = [sc.to(cuda), sh.to(cuda)]                   # (src() keeps nothing alive)
    s0 = src(ad, t[0], t[1], relu=1, up=1) if lazy else src(ad, up=1)
    L.check(L.lib().uwm_op_conv(C.byref(s0), None, P(wp), 16, kpad, 3, 3, 1, 1, n, 16, None, P(out), P(stats), cfg, stream()))
    torch.cuda.synchronize()
    got = nchw(out.cpu(), 16).double()
    err = float((got - y).abs().max())
    assert err < 2e-5 * max(1.0, float(y.abs().max())), err
    assert torch.allclose(stats[:16].cpu(), y.sum((0, 2, 3)), rtol=1e-5, atol=1e-3)
    assert torch.allclose(stats[16:].cpu(), (y * y).sum((0, 2, 3)), rtol=1e-5, atol=1e-3)
    # the Winograd route on the same operands (what this kernel replaces) agrees to fp32 rounding
    out2 = torch.empty_like(out)
    L.check(L.lib().uwm_op_conv(C.byref(s0), None, P(wp), 16, kpad, 3, 3, 1, 1, n, 16, None, P(out2), None, 300, stream()))
    assert float((out2 - out).abs().max()) < 5e-5 * max(1.0, float(y.abs().max()))


def test_conv_winograd_error_vs_fp64(cuda):
    """Winograd's transforms cost a little accuracy; measured against an fp64 convolution the error must stay
    within 4x that of the direct fp32 kernel (and far inside the 1e-3 logit budget)."""
    L = lib()
    g = torch.Generator().manual_seed(11)
    n, cin, cout, h, w = 2, 128, 64, 32, 32
    x = torch.randn(n, cin, h, w, generator=g)
    wt = torch.randn(cout, cin, 3, 3, generator=g) * (1.0 / (cin * 9) ** 0.5)
    ref = F.conv2d(x.double(), wt.double(), None, 1, 1)
    xd = nhwc(x).to(cuda)
    wp, kpad = pack_w(wt)
    wp = wp.to(cuda)
    errs = {}
    for name, cfg in (("wino", 300), ("direct", 164)):
        y = torch.empty(n, h, w, cout, device=cuda)
        s0 = src(xd)
        L.check(L.lib().uwm_op_conv(C.byref(s0), None, P(wp), cout, kpad, 3, 3, 1, 1, n, cout, None, P(y), None, cfg, stream()))
        torch.cuda.synchronize()
        errs[name] = float((nchw(y.cpu()).double() - ref).abs().max())
    assert errs["wino"] < 4 * errs["direct"] + 1e-6 and errs["wino"] < 2e-5, errs


@pytest.mark.parametrize("cout", [16, 1, 32])
def test_conv_patch16(cuda, cout):
    """16-channel-input kernel (whole K=144 in LDS): decoder block 4 conv2 / head / dcat-dgrad shapes;
    40x24 pixels -> partial 16x16 tiles; lazy BatchNorm+ReLU input."""
    _conv_case(cuda, 2, 16, cout, 40, 24, 3, 1, 1, cfg=200)
    _conv_case(cuda, 1, 16, cout, 16, 16, 3, 1, 1, cfg=200, lazy=True, seed=4)


@pytest.mark.parametrize("shape", [
    (2, 3, 64, 64, 64, 7, 2, 3),      # stem: Cin 3->4 pad, 7x7 s2, K=196->224
    (2, 64, 128, 32, 32, 3, 2, 1),    # layer2.0.conv1: stride 2
    (2, 64, 128, 32, 32, 1, 2, 0),    # downsample 1x1 s2
    (1, 16, 16, 32, 64, 3, 1, 1),     # decoder block 4 conv2: Cin=16 (two taps per 32-chunk)
    (1, 16, 1, 32, 32, 3, 1, 1),      # head: Cout 1 -> padded 4
    (3, 256, 512, 8, 8, 3, 1, 1),     # deep stage, ragged M (192 pixels)
    (1, 128, 32, 16, 48, 3, 1, 1),    # Cout 32 tile
])
def test_conv_shapes(cuda, shape):
    n, cin, cout, h, w, k, s, p = shape
    _conv_case(cuda, n, cin, cout, h, w, k, s, p)


@pytest.mark.parametrize("shape", [
    (2, 24, 144, 16, 24),      # EfficientNet expand: K = 24 (padded to one 32-float chunk), Cout = 144
    (1, 144, 24, 24, 40),      # project: Cout 24
    (3, 48, 12, 8, 8),         # SE-sized channel counts, ragged M
    (1, 272, 1632, 8, 16),     # deep expand
    (1, 1632, 272, 8, 8),      # deep project conv: K = 1632
])
def test_conv1x1_shapes(cuda, shape):
    """1x1 convs of the MBConv / Bottleneck blocks on the flattened implicit GEMM: plain and lazy sources, statistics."""
    n, cin, cout, h, w = shape
    _conv_case(cuda, n, cin, cout, h, w, 1, 1, 0)
    _conv_case(cuda, n, cin, cout, h, w, 1, 1, 0, lazy=True, seed=2)


@pytest.mark.parametrize("shape", [
    (2, 64, 256, 16, 24),      # Bottleneck conv3 of layer1: K = 64 (two chunks per tile), two 128-channel tiles, 6 pixel tiles
    (1, 256, 64, 16, 16),      # conv1: K = 256, one 64-channel tile
    (3, 128, 96, 8, 12),       # ragged M (288 = 2.25 tiles), Cout 96 inside a 128-channel tile
    (2, 96, 38, 16, 16),       # 38 real outputs in 40 padded channels: the pad must come out as exact zeros
    (1, 32, 64, 16, 8),        # a single K chunk, M = 128 exactly
    (5, 160, 192, 8, 8),       # MBConv-sized: K = 160 (five chunks), 2 + 1 channel tiles on the 64-wide config
    (2, 24, 144, 16, 24),      # K = 24: one PARTIAL chunk (clamped duplicate units of X against W's zero padding)
    (1, 144, 40, 16, 24),      # K = 144 = 4.5 chunks, 40 outputs
    (2, 56, 336, 16, 16),      # K = 56
])
@pytest.mark.parametrize("cfg", [-1, 864, 928])
def test_conv_gemm_1x1(cuda, shape, cfg):
    """conv_gemm.hip: 1x1 / stride-1 convs with Cin % 32 == 0 as a persistent LDS-DMA GEMM (a workgroup walks several tiles, the
    statistics of one channel tile leave once, the lazy BatchNorm + ReLU is applied to the fragments after ds_read); auto route and
    both tile widths, plain and lazy sources, BatchNorm statistics, zero padded channels."""
    n, cin, cout, h, w = shape
    _conv_case(cuda, n, cin, cout, h, w, 1, 1, 0, cfg=cfg)
    _conv_case(cuda, n, cin, cout, h, w, 1, 1, 0, cfg=cfg, lazy=True, seed=5)


@pytest.mark.parametrize("cin,cout,n,h,w", [(16, 1, 2, 32, 48), (16, 3, 1, 20, 36), (8, 1, 1, 9, 17), (32, 4, 2, 16, 16), (16, 2, 1, 64, 300)])
def test_conv_head_streaming_kernel(cuda, cin, cout, n, h, w):
    """conv_head.hip (3x3 segmentation head, <= 4 classes): forced (cfg 500) and auto-routed, plain and lazy sources, bias,
    zero padding applied after the producer's activation, odd sizes (ragged 4-row bands)."""
    L = lib()
    g = torch.Generator().manual_seed(3)
    x = torch.randn(n, cin, h, w, generator=g)
    wt = torch.randn(cout, cin, 3, 3, generator=g) * 0.1
    bias = torch.randn(cout, generator=g)
    scale = torch.rand(cin, generator=g) + 0.5; scale[::3] *= -1
    shift = torch.randn(cin, generator=g) * 0.3
    wp, kpad = pack_w(wt)
    bp = torch.zeros(4); bp[:cout] = bias
    xd, wd, bd, scd, shd = nhwc(x).to(cuda), wp.to(cuda), bp.to(cuda), scale.to(cuda), shift.to(cuda)
    for lazy in (False, True):
        xin = torch.relu(x * scale[:, None, None] + shift[:, None, None]) if lazy else x
        ref = F.conv2d(xin, wt, bias, 1, 1)
        for cfg in (500, -1):
            y = torch.full((n, h, w, 4), float("nan"), device=cuda)
            s0 = src(xd, scd if lazy else None, shd if lazy else None, relu=1 if lazy else 0)
            L.check(L.lib().uwm_op_conv(C.byref(s0), None, P(wd), cout, kpad, 3, 3, 1, 1, n, 4, P(bd), P(y), None, cfg, stream()))
            torch.cuda.synchronize()
            assert (nchw(y.cpu(), cout) - ref).abs().max() < 2e-5 * max(1.0, float(ref.abs().max()))
            assert (y[..., cout:] == 0).all()


def test_conv_lazy_bn_relu_prologue(cuda):
    # consumer-side BatchNorm-apply + ReLU (negative scales included) and zero padding AFTER it
    _conv_case(cuda, 2, 64, 64, 16, 16, 3, 1, 1, lazy=True)
    _conv_case(cuda, 2, 16, 32, 16, 16, 3, 1, 1, lazy=True, seed=3)


def test_conv_upsample_concat(cuda):
    """decoder block conv1: cat(nearest_x2(d), skip) fused into the gather; both sources lazy."""
    L = lib()
    g = torch.Generator().manual_seed(1)
    n, c0, c1, cout, h, w = 2, 64, 32, 32, 8, 12
    d = torch.randn(n, c0, h, w, generator=g)
    sk = torch.randn(n, c1, 2 * h, 2 * w, generator=g)
    sc0, sh0 = torch.rand(c0, generator=g) + 0.5, torch.randn(c0, generator=g) * 0.2
    sc1, sh1 = torch.rand(c1, generator=g) + 0.5, torch.randn(c1, generator=g) * 0.2
    wt = torch.randn(cout, c0 + c1, 3, 3, generator=g) * 0.05
    a0 = torch.relu(d * sc0[:, None, None] + sh0[:, None, None])
    a1 = torch.relu(sk * sc1[:, None, None] + sh1[:, None, None])
    ref = F.conv2d(torch.cat([F.interpolate(a0, scale_factor=2, mode="nearest"), a1], 1), wt, None, 1, 1)
    dd, skd = nhwc(d).to(cuda), nhwc(sk).to(cuda)
    wp, kpad = pack_w(wt)
    wp = wp.to(cuda)
    t = [sc0.to(cuda), sh0.to(cuda), sc1.to(cuda), sh1.to(cuda)]
    y = torch.empty(n, 2 * h, 2 * w, cout, device=cuda)
    s0, s1 = src(dd, t[0], t[1], relu=1, up=1), src(skd, t[2], t[3], relu=1)
    L.check(L.lib().uwm_op_conv(C.byref(s0), C.byref(s1), P(wp), cout, kpad, 3, 3, 1, 1, n, cout, None, P(y), None, -1,
                                stream()))
    torch.cuda.synchronize()
    assert (nchw(y.cpu()) - ref).abs().max() < 2e-5 * float(ref.abs().max())


@pytest.mark.parametrize("shape", [
    (2, 64, 64, 16, 24, 3, 1, 1),
    (2, 64, 128, 32, 32, 3, 2, 1),
    (2, 64, 128, 32, 32, 1, 2, 0),
    (1, 16, 1, 32, 32, 3, 1, 1),      # head dgrad: dy has 1 -> 4 padded channels (conv_head_dgrad streaming kernel)
    (2, 32, 3, 18, 20, 3, 1, 1),      # head dgrad, 3 classes -> 32 channels, ragged bands
    (1, 8, 4, 9, 33, 3, 1, 1),        # head dgrad, 4 classes -> 8 channels 
    (3, 16, 1, 64, 70, 3, 1, 1),      # head layer over several images / bands / column blocks
    (1, 96, 32, 16, 16, 3, 1, 1),     # concat input (Ctot=96): dgrad output is the full dcat
    (2, 32, 16, 24, 40, 3, 1, 1),     # wgrad_patch<16>, partial 8x16 tiles
    (1, 64, 128, 16, 16, 3, 1, 1),    # wgrad_patch<64> with two output-channel tiles
    (3, 64, 32, 8, 16, 3, 1, 1),      # wgrad_patch<32>, one tile per image
    (1, 32, 16, 32, 48, 3, 1, 1),     # dgrad runs conv_patch16<32> (dy has 16 channels)
    (2, 16, 16, 16, 32, 3, 1, 1),     # dgrad runs conv_patch16<16>
    (2, 24, 144, 16, 24, 1, 1, 0),    # 1x1 stride 1 (EfficientNet expand): dgrad through conv_1x1 (K = 144, 24 outputs)
    (1, 144, 40, 8, 24, 1, 1, 0),     # 1x1 project: dgrad K = 40 (2.5 k-steps), 144 outputs, ragged M
    (2, 56, 336, 8, 16, 1, 1, 0),     # 1x1 expand with Kpad = 64: wgrad on the 128x64 tile (Kpad = 32 above: 128x32)
    (1, 160, 192, 8, 8, 1, 1, 0),     # Kpad = 160: three 64-column tiles instead of two 128-column ones
    (2, 48, 24, 16, 16, 1, 1, 0),     # 1x1 project with 24 outputs, Kpad = 64: wgrad on the 32x64 tile
    (2, 64, 128, 16, 16, 1, 1, 0),    # 1x1 whose dgrad (K = 128 dY channels) runs on conv_gemm.hip with addend + mask
    (1, 256, 64, 16, 24, 1, 1, 0),    # Bottleneck conv1: dgrad K = 64, 256 outputs (two channel tiles)
    (1, 144, 32, 8, 24, 1, 1, 0),     # Kpad = 160 under 32 outputs: the 32x64 / 32x128 tiles
    (1, 96, 24, 8, 8, 1, 1, 0),       # Kpad = 96
])
@pytest.mark.parametrize("force_igemm", [0, 1, 2, 3])
def test_dgrad_and_wgrad(cuda, shape, force_igemm):
    """force_igemm: 0 = auto (Winograd dgrad / Winograd-domain wgrad where applicable), 1 = flattened implicit GEMM
    wgrad with Winograd off (direct dgrad kernels), 2 = patch wgrad with Winograd on, 3 = auto with the 8-wave
    Winograd variant preferred (dgrad through conv_wino8 where the shape allows)."""
    L = lib()
    L.lib().uwm_set_winograd({0: 1, 1: 0, 2: 1, 3: 2}[force_igemm])
    try:
        _dgrad_and_wgrad(cuda, shape, 0 if force_igemm == 3 else force_igemm)
    finally:
        L.lib().uwm_set_winograd(1)


def _dgrad_and_wgrad(cuda, shape, force_igemm):
    L = lib()
    n, cin, cout, h, w, k, s, p = shape
    g = torch.Generator().manual_seed(5)
    x = torch.randn(n, cin, h, w, generator=g, requires_grad=True)
    wt = (torch.randn(cout, cin, k, k, generator=g) * 0.05).requires_grad_()
    y = F.conv2d(x, wt, None, s, p)
    dy = torch.randn(y.shape, generator=g)
    y.backward(dy)
    ho, wo = y.shape[-2:]
    coutp = rup(cout, 4)
    wp, kpad = pack_w(wt.detach())
    wp = wp.to(cuda)
    kpadd = rup(k * k * coutp, 32)
    wd = torch.empty(cin, kpadd, device=cuda)
    L.check(L.lib().uwm_op_pack_dgrad(P(wp), cout, kpad, k * k, cin, P(wd), kpadd, coutp, stream()))
    dyd = nhwc(dy).to(cuda)
    addend = torch.randn(n, h, w, cin, generator=g)
    maskt = torch.randn(n, h, w, cin, generator=g)
    add_d, mask_d = addend.to(cuda), maskt.to(cuda)
    dx = torch.empty(n, h, w, cin, device=cuda)
    L.check(L.lib().uwm_op_dgrad(P(dyd), n, ho, wo, coutp, P(wd), cin, kpadd, k, k, s, p, h, w, P(add_d), P(mask_d), None,
                                 None, P(dx), stream()))
    dw = torch.zeros(cout, kpad, device=cuda)
    xd = nhwc(x.detach()).to(cuda)
    s0 = src(xd)
    L.check(L.lib().uwm_op_wgrad(C.byref(s0), None, P(dyd), n, ho, wo, coutp, cout, kpad, k, k, s, p, P(dw), force_igemm, stream()))
    torch.cuda.synchronize()
    ref_dx = (x.grad.permute(0, 2, 3, 1) + addend) * (maskt > 0)
    assert (dx.cpu() - ref_dx).abs().max() < 3e-5 * max(1.0, float(ref_dx.abs().max()))
    got_dw = unpack_w(dw.cpu(), cout, cin, k, k)
    assert (got_dw - wt.grad).abs().max() < 3e-5 * max(1.0, float(wt.grad.abs().max()))
    # padded K region of dW must stay exactly zero (it aliases arena padding)
    assert (dw[:, k * k * rup(cin, 4):] == 0).all()


@pytest.mark.parametrize("cout,n,h,w", [(16, 2, 16, 64), (16, 3, 40, 32), (1, 2, 16, 64), (3, 2, 24, 96), (16, 4, 128, 128)])
def test_wgrad_c16_and_head(cuda, cout, n, h, w):
    """wgrad_c16.hip: the weight gradient of the 16-channel full-resolution layers (16 -> 16: pixels as the MFMA reduction,
    9 accumulators) and of the segmentation head (16 -> 1 | 3 classes: taps as the MFMA N dimension), lazily normalised
    + ReLU-ed input, many tiles per workgroup (128x128 x 4 images), image borders; deterministic: two runs are bit-identical;
    dW is accumulated (+=) and its padded K columns stay zero."""
    L = lib()
    g = torch.Generator().manual_seed(11 + cout)
    cin = 16
    xr = torch.randn(n, cin, h, w, generator=g)
    sc = torch.rand(cin, generator=g) + 0.5; sc[::5] *= -1
    sh = torch.randn(cin, generator=g) * 0.3
    x = torch.relu(xr * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1))
    wt = (torch.randn(cout, cin, 3, 3, generator=g) * 0.05).requires_grad_()
    y = F.conv2d(x, wt, None, 1, 1)
    dy = torch.randn(y.shape, generator=g)
    y.backward(dy)
    coutp = rup(cout, 4)
    kpad = rup(9 * cin, 32)
    xd, dyd = nhwc(xr).to(cuda), nhwc(dy).to(cuda)
    scd, shd = sc.to(cuda), sh.to(cuda)
    s0 = src(xd, scd, shd, relu=1)
    runs = []
    for _ in range(2):
        dw = torch.full((cout, kpad), 0.25, device=cuda); dw[:, 9 * cin:] = 0
        L.check(L.lib().uwm_op_wgrad(C.byref(s0), None, P(dyd), n, h, w, coutp, cout, kpad, 3, 3, 1, 1, P(dw), 0, stream()))
        torch.cuda.synchronize()
        runs.append(dw.clone())
    assert torch.equal(runs[0], runs[1])                       # no atomics
    got = unpack_w(runs[0].cpu(), cout, cin, 3, 3) - 0.25
    ref = wt.grad
    assert (got - ref).abs().max() < 3e-5 * max(1.0, float(ref.abs().max())), float((got - ref).abs().max())
    assert (runs[0][:, 9 * cin:] == 0).all()


@pytest.mark.parametrize("n,h,w", [(4, 128, 128), (1, 8, 32), (2, 24, 96)])
def test_wgrad_c16_f16x3(cuda, n, h, w):
    """wgrad_c16.hip's fp16x3 kernel (force 6: WgradArgs::prec == 2 — decoder block 4 conv2 in the f16x3_all modes): split products
    on v_mfma_f32_16x16x32_f16, both operands through transposing LDS reads, persistent double-buffered stages.  Lazy BatchNorm +
    ReLU input (negative scales included), dY ~ 1e-6 (scaled through max|dY|); against fp64 autograd within 4x the exact-fp32
    kernel's own error, bit-identical between two launches, dW accumulated (+=), padded K columns left at zero."""
    L = lib()
    g = torch.Generator().manual_seed(71 + h)
    cin = cout = 16
    xr = torch.randn(n, cin, h, w, generator=g) * 2.0
    sc = torch.rand(cin, generator=g) + 0.5; sc[::5] *= -1
    sh = torch.randn(cin, generator=g) * 0.3
    x = torch.relu(xr * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1)).double()
    wt = (torch.randn(cout, cin, 3, 3, generator=g) * 0.05).double().requires_grad_()
    y = F.conv2d(x, wt, None, 1, 1)
    dy = torch.randn(y.shape, generator=g) * 1e-6
    y.backward(dy.double())
    kpad = rup(9 * cin, 32)
    xd, dyd = nhwc(xr).to(cuda), nhwc(dy).to(cuda)
    scd, shd = sc.to(cuda), sh.to(cuda)
    s0 = src(xd, scd, shd, relu=1)
    outs = {}
    for name, force in (("f32", 0), ("f16x3", 6), ("again", 6)):
        dw = torch.full((cout, kpad), 0.25, device=cuda); dw[:, 9 * cin:] = 0
        L.check(L.lib().uwm_op_wgrad(C.byref(s0), None, P(dyd), n, h, w, cout, cout, kpad, 3, 3, 1, 1, P(dw), force, stream()))
        torch.cuda.synchronize()
        outs[name] = dw.cpu()
    assert torch.equal(outs["f16x3"], outs["again"])
    assert not torch.equal(outs["f16x3"], outs["f32"])          # (the fp16x3 kernel did run)
    ref = wt.grad
    err = {k: float(((unpack_w(outs[k], cout, cin, 3, 3).double() - 0.25) - ref).abs().max() / ref.abs().max()) for k in ("f32", "f16x3")}
    # (dW = 0.25 + a gradient of ~1e-5: both kernels sit at the fp32 rounding of that sum)
    assert err["f16x3"] < 4 * err["f32"] + 1e-6, err
    assert (outs["f16x3"][:, 9 * cin:] == 0).all()
    # the same without the 0.25 offset: the gradient's own digits
    dw = torch.zeros(cout, kpad, device=cuda)
    L.check(L.lib().uwm_op_wgrad(C.byref(s0), None, P(dyd), n, h, w, cout, cout, kpad, 3, 3, 1, 1, P(dw), 6, stream()))
    torch.cuda.synchronize()
    e0 = float((unpack_w(dw.cpu(), cout, cin, 3, 3).double() - ref).abs().max() / ref.abs().max())
    assert e0 < 1e-5, e0


@pytest.mark.parametrize("c0,c1,cout", [(32, 16, 16), (64, 32, 32), (32, 32, 64),
                                        (64, 48, 32), (32, 24, 64), (64, 56, 128)])      # channel tails of the EfficientNet decoder concats (Ctot % 32 != 0)
def test_wgrad_lazy_upsample_concat(cuda, c0, c1, cout):
    L = lib()
    g = torch.Generator().manual_seed(7)
    n, h, w = 2, 8, 8
    d = torch.randn(n, c0, h, w, generator=g)
    sk = torch.randn(n, c1, 2 * h, 2 * w, generator=g)
    sc0, sh0 = torch.rand(c0, generator=g) + 0.5, torch.randn(c0, generator=g) * 0.2
    a0 = torch.relu(d * sc0[:, None, None] + sh0[:, None, None])
    xin = torch.cat([F.interpolate(a0, scale_factor=2, mode="nearest"), sk], 1)
    wt = (torch.randn(cout, c0 + c1, 3, 3, generator=g) * 0.05).requires_grad_()
    y = F.conv2d(xin, wt, None, 1, 1)
    dy = torch.randn(y.shape, generator=g)
    y.backward(dy)
    dd, skd, dyd = nhwc(d).to(cuda), nhwc(sk).to(cuda), nhwc(dy).to(cuda)
    t = [sc0.to(cuda), sh0.to(cuda)]
    kpad = rup(9 * (c0 + c1), 32)
    dw = torch.zeros(cout, kpad, device=cuda)
    s0, s1 = src(dd, t[0], t[1], relu=1, up=1), src(skd)
    L.check(L.lib().uwm_op_wgrad(C.byref(s0), C.byref(s1), P(dyd), n, 2 * h, 2 * w, cout, cout, kpad, 3, 3, 1, 1, P(dw),
                                 0, stream()))
    torch.cuda.synchronize()
    got = unpack_w(dw.cpu(), cout, c0 + c1, 3, 3)
    assert (got - wt.grad).abs().max() < 3e-5 * max(1.0, float(wt.grad.abs().max()))
    assert (dw.cpu()[:, 9 * (c0 + c1):] == 0).all()           # K padding of dW stays exactly zero


@pytest.mark.parametrize("n,h,w", [(2, 64, 64), (1, 128, 192), (3, 32, 64)])
def test_wgrad_stem_compact_columns(cuda, n, h, w):
    """wgrad_stem.hip: weight gradient of the 7x7 / stride-2 / pad-3 ResNet stem (3 input channels stored as 4) with compact GEMM
    columns (147 real of 160) against autograd, bit-identical between two launches, pad channel / pad K of dW left untouched,
    and equal (to rounding) to the flattened implicit GEMM it replaces."""
    L = lib()
    g = torch.Generator().manual_seed(51 + h)
    x = torch.randn(n, 3, h, w, generator=g)
    wt = (torch.randn(64, 3, 7, 7, generator=g) * 0.05).requires_grad_()
    y = F.conv2d(x, wt, None, 2, 3)
    dy = torch.randn(y.shape, generator=g)
    y.backward(dy)
    xd, dyd = nhwc(x).to(cuda), nhwc(dy).to(cuda)
    kpad = rup(49 * 4, 32)
    s0 = src(xd)
    outs = []
    for force in (0, 0, 1):
        dw = torch.zeros(64, kpad, device=cuda)
        L.check(L.lib().uwm_op_wgrad(C.byref(s0), None, P(dyd), n, h // 2, w // 2, 64, 64, kpad, 7, 7, 2, 3, P(dw), force, stream()))
        torch.cuda.synchronize()
        outs.append(dw.cpu())
    assert torch.equal(outs[0], outs[1])
    got = unpack_w(outs[0], 64, 3, 7, 7)
    tol = 3e-5 * max(1.0, float(wt.grad.abs().max()))
    assert (got - wt.grad).abs().max() < tol
    assert (outs[0] - outs[2]).abs().max() < tol
    full = outs[0][:, :196].reshape(64, 49, 4)
    assert (full[..., 3] == 0).all() and (outs[0][:, 196:] == 0).all()


@pytest.mark.parametrize("n,h,w", [(2, 64, 64), (1, 128, 192), (3, 32, 64)])
def test_wgrad_stem_f16x3(cuda, n, h, w):
    """wgrad_stem.hip's fp16x3 kernel (force 6: WgradArgs::prec == 2 — what the model takes for encoder.conv1 in the f16x3_all modes):
    split products on v_mfma_f32_16x16x32_f16, both operands through transposing LDS reads, columns in dW's own tap*4 + c layout.
    dY as tiny as a real Dice gradient (scaled through max|dY|); against fp64 autograd within 4x the exact-fp32 kernel's own error,
    bit-identical between two launches, pad channel / pad K of dW left at zero."""
    L = lib()
    g = torch.Generator().manual_seed(61 + h)
    x = torch.randn(n, 3, h, w, generator=g) * 1.5
    wt = (torch.randn(64, 3, 7, 7, generator=g) * 0.05).double().requires_grad_()
    y = F.conv2d(x.double(), wt, None, 2, 3)
    dy = torch.randn(y.shape, generator=g) * 1e-6
    y.backward(dy.double())
    xd, dyd = nhwc(x).to(cuda), nhwc(dy).to(cuda)
    kpad = rup(49 * 4, 32)
    s0 = src(xd)
    outs = {}
    for name, force in (("f32", 0), ("f16x3", 6), ("again", 6)):
        dw = torch.zeros(64, kpad, device=cuda)
        L.check(L.lib().uwm_op_wgrad(C.byref(s0), None, P(dyd), n, h // 2, w // 2, 64, 64, kpad, 7, 7, 2, 3, P(dw), force, stream()))
        torch.cuda.synchronize()
        outs[name] = dw.cpu()
    assert torch.equal(outs["f16x3"], outs["again"])
    ref = wt.grad
    err = {k: float((unpack_w(outs[k], 64, 3, 7, 7).double() - ref).abs().max() / ref.abs().max()) for k in ("f32", "f16x3")}
    assert err["f16x3"] < 4 * err["f32"] + 1e-7 and err["f16x3"] < 1e-5, err
    assert not torch.equal(outs["f16x3"], outs["f32"])          # (the fp16x3 kernel did run)
    full = outs["f16x3"][:, :196].reshape(64, 49, 4)
    assert (full[..., 3] == 0).all() and (outs["f16x3"][:, 196:] == 0).all()


@pytest.mark.parametrize("n,cin,cout,h,w", [(2, 64, 128, 16, 16), (1, 256, 64, 32, 32), (4, 96, 40, 16, 16), (1, 160, 192, 32, 32), (2, 32, 32, 16, 8),
                                            (2, 56, 336, 16, 16), (1, 144, 32, 32, 32), (4, 24, 144, 16, 16)])      # Cin % 32 != 0: dW pad columns stay zero
@pytest.mark.parametrize("lazy", [False, True])
def test_wgrad_gemm_1x1(cuda, n, cin, cout, h, w, lazy):
    """wgrad_gemm.hip: weight gradient of the 1x1 / stride-1 convs (Cin % 32 == 0) as an LDS-DMA GEMM over the pixels with per-split
    partial images added in a fixed order: against autograd (plain and lazily normalised + ReLU-ed input), bit-identical between two
    launches, partial channel tiles (Cout 40 in a 64 tile, 192 = 128 + 64, Cin 96 / 160 in 128 tiles)."""
    L = lib()
    g = torch.Generator().manual_seed(41 + cin)
    x = torch.randn(n, cin, h, w, generator=g)
    sc, sh = torch.rand(cin, generator=g) + 0.5, torch.randn(cin, generator=g) * 0.3
    xin = torch.relu(x * sc[:, None, None] + sh[:, None, None]) if lazy else x
    wt = (torch.randn(cout, cin, 1, 1, generator=g) * 0.05).requires_grad_()
    y = F.conv2d(xin, wt)
    dy = torch.randn(y.shape, generator=g)
    y.backward(dy)
    coutp = rup(cout, 4)
    xd, dyd = nhwc(x).to(cuda), nhwc(dy, coutp).to(cuda)
    t = [sc.to(cuda), sh.to(cuda)]
    s0 = src(xd, t[0], t[1], relu=1) if lazy else src(xd)
    kpad = rup(cin, 32)
    outs = []
    for _ in range(2):
        dw = torch.zeros(cout, kpad, device=cuda)
        L.check(L.lib().uwm_op_wgrad(C.byref(s0), None, P(dyd), n, h, w, coutp, cout, kpad, 1, 1, 1, 0, P(dw), 4, stream()))
        torch.cuda.synchronize()
        outs.append(dw.cpu())
    assert torch.equal(outs[0], outs[1])
    got = unpack_w(outs[0], cout, cin, 1, 1)
    assert (got - wt.grad).abs().max() < 3e-5 * max(1.0, float(wt.grad.abs().max()))
    assert (outs[0][:, cin:] == 0).all()
    # accumulate semantics: a second launch into the same buffer doubles it
    dw = outs[0].to(cuda).clone()
    L.check(L.lib().uwm_op_wgrad(C.byref(s0), None, P(dyd), n, h, w, coutp, cout, kpad, 1, 1, 1, 0, P(dw), 4, stream()))
    assert (dw.cpu() - 2 * outs[0]).abs().max() < 1e-5 * max(1.0, float(outs[0].abs().max()))


@pytest.mark.parametrize("n,hs,ws", [(2, 8, 16), (3, 12, 32)])
def test_wgrad_up2_subpixel(cuda, n, hs, ws):
    """decoder block 4 conv1 weight gradient (32 upsampled channels -> 16) in sub-pixel form (conv_up2.hip: 16 class products
    over the low-resolution tensor, folded into the nine taps by the reduce kernel) against autograd, and bit-identical
    between two launches (fixed-order partial sums)."""
    L = lib()
    g = torch.Generator().manual_seed(31 + hs)
    d = torch.randn(n, 32, hs, ws, generator=g)
    sc0, sh0 = torch.rand(32, generator=g) + 0.5, torch.randn(32, generator=g) * 0.2
    a0 = torch.relu(d * sc0[:, None, None] + sh0[:, None, None])
    wt = (torch.randn(16, 32, 3, 3, generator=g) * 0.05).requires_grad_()
    y = F.conv2d(F.interpolate(a0, scale_factor=2, mode="nearest"), wt, None, 1, 1)
    dy = torch.randn(y.shape, generator=g)
    y.backward(dy)
    dd, dyd = nhwc(d).to(cuda), nhwc(dy).to(cuda)
    kpad = rup(9 * 32, 32)
    t = [sc0.to(cuda), sh0.to(cuda)]                 # (src() keeps nothing alive)
    s0 = src(dd, t[0], t[1], relu=1, up=1)
    outs = []
    for _ in range(2):
        dw = torch.zeros(16, kpad, device=cuda)
        L.check(L.lib().uwm_op_wgrad(C.byref(s0), None, P(dyd), n, 2 * hs, 2 * ws, 16, 16, kpad, 3, 3, 1, 1, P(dw), 0, stream()))
        torch.cuda.synchronize()
        outs.append(dw.cpu())
    assert torch.equal(outs[0], outs[1])
    got = unpack_w(outs[0], 16, 32, 3, 3)
    assert (got - wt.grad).abs().max() < 3e-5 * max(1.0, float(wt.grad.abs().max()))
    # the generic route (force_igemm = 1) on the same operands
    dw2 = torch.zeros(16, kpad, device=cuda)
    L.check(L.lib().uwm_op_wgrad(C.byref(s0), None, P(dyd), n, 2 * hs, 2 * ws, 16, 16, kpad, 3, 3, 1, 1, P(dw2), 1, stream()))
    assert (dw2.cpu() - outs[0]).abs().max() < 3e-5 * max(1.0, float(wt.grad.abs().max()))


@pytest.mark.parametrize("n,hs,ws", [(2, 16, 32), (1, 2, 32), (3, 8, 64)])
def test_wgrad_up2_f16x3(cuda, n, hs, ws):
    """decoder block 4 conv1 weight gradient on conv_up2_f16.hip's fp16x3 kernel (force 6: WgradArgs::prec == 2): the sixteen class
    products with the low-resolution pixels as the MFMA k dimension, transposing LDS reads, the fp32 kernel's partial layout and
    reduce.  Lazy input with negative scales, dY ~ 1e-6; against fp64 autograd within 4x the exact-fp32 sub-pixel kernel's own error,
    bit-identical between two launches."""
    L = lib()
    g = torch.Generator().manual_seed(33 + hs)
    d = torch.randn(n, 32, hs, ws, generator=g) * 2.0
    sc0, sh0 = torch.rand(32, generator=g) + 0.5, torch.randn(32, generator=g) * 0.2
    sc0[::6] *= -1
    a0 = torch.relu(d * sc0[:, None, None] + sh0[:, None, None]).double()
    wt = (torch.randn(16, 32, 3, 3, generator=g) * 0.05).double().requires_grad_()
    y = F.conv2d(F.interpolate(a0, scale_factor=2, mode="nearest"), wt, None, 1, 1)
    dy = torch.randn(y.shape, generator=g) * 1e-6
    y.backward(dy.double())
    dd, dyd = nhwc(d).to(cuda), nhwc(dy).to(cuda)
    kpad = rup(9 * 32, 32)
    t = [sc0.to(cuda), sh0.to(cuda)]
    s0 = src(dd, t[0], t[1], relu=1, up=1)
    outs = {}
    for name, force in (("f32", 0), ("f16x3", 6), ("again", 6)):
        dw = torch.zeros(16, kpad, device=cuda)
        L.check(L.lib().uwm_op_wgrad(C.byref(s0), None, P(dyd), n, 2 * hs, 2 * ws, 16, 16, kpad, 3, 3, 1, 1, P(dw), force, stream()))
        torch.cuda.synchronize()
        outs[name] = dw.cpu()
    assert torch.equal(outs["f16x3"], outs["again"])
    assert not torch.equal(outs["f16x3"], outs["f32"])          # (the fp16x3 kernel did run)
    ref = wt.grad
    err = {k: float((unpack_w(outs[k], 16, 32, 3, 3).double() - ref).abs().max() / ref.abs().max()) for k in ("f32", "f16x3")}
    assert err["f16x3"] < 4 * err["f32"] + 1e-7 and err["f16x3"] < 1e-5, err


@pytest.mark.parametrize("shape", [
    (2, 64, 128, 32, 32, 3, 2, 1),    # layer2.0.conv1: 3x3 stride 2 (128 x 128 tiles)
    (2, 64, 128, 32, 32, 1, 2, 0),    # downsample 1x1 stride 2 (Kpad 64: the 128 x 64 tile)
    (1, 128, 256, 16, 24, 3, 2, 1),   # layer3.0.conv1, a ragged last pixel step
    (3, 32, 64, 16, 16, 3, 2, 1),     # 64 output rows: the 64 x 128 tile
])
def test_wgrad_igemm_f16x3_stride2_layers(cuda, shape):
    """wgrad_igemm.hip's fp16x3 form (force 7: WgradArgs::prec == 2 on the flattened implicit GEMM — the stride-2 layers' weight
    gradients in the f16x3_all modes): a 32-pixel step = one v_mfma_f32_16x16x32_f16 k-step, operands split while staged, transposing
    LDS reads.  Lazy BatchNorm + ReLU input with negative scales, dY ~ 1e-6 (scaled through max|dY|); against fp64 autograd within 4x
    the exact-fp32 kernel's own error; bit-identical between two launches."""
    L = lib()
    n, cin, cout, h, w, k, st, p = shape
    g = torch.Generator().manual_seed(17 + cin)
    xr = torch.randn(n, cin, h, w, generator=g) * 2.0
    sc = torch.rand(cin, generator=g) + 0.5; sc[::5] *= -1
    sh = torch.randn(cin, generator=g) * 0.2
    x = torch.relu(xr * sc[:, None, None] + sh[:, None, None]).double()
    wt = (torch.randn(cout, cin, k, k, generator=g) * 0.05).double().requires_grad_()
    y = F.conv2d(x, wt, None, st, p)
    ho, wo = y.shape[-2:]
    dy = torch.randn(y.shape, generator=g) * 1e-6
    y.backward(dy.double())
    kpad = rup(k * k * cin, 32)
    xd, dyd = nhwc(xr).to(cuda), nhwc(dy).to(cuda)
    scd, shd = sc.to(cuda), sh.to(cuda)
    s0 = src(xd, scd, shd, relu=1)
    outs = {}
    for name, force in (("f32", 1), ("f16x3", 7), ("again", 7)):
        dw = torch.zeros(cout, kpad, device=cuda)
        L.check(L.lib().uwm_op_wgrad(C.byref(s0), None, P(dyd), n, ho, wo, cout, cout, kpad, k, k, st, p, P(dw), force, stream()))
        torch.cuda.synchronize()
        outs[name] = dw.cpu()
    assert torch.equal(outs["f16x3"], outs["again"])
    assert not torch.equal(outs["f16x3"], outs["f32"])          # (the fp16x3 kernel did run)
    ref = wt.grad
    err = {kk: float((unpack_w(outs[kk], cout, cin, k, k).double() - ref).abs().max() / ref.abs().max()) for kk in ("f32", "f16x3")}
    assert err["f16x3"] < 4 * err["f32"] + 1e-7 and err["f16x3"] < 1e-5, err
    assert (outs["f16x3"][:, k * k * cin:] == 0).all()


def test_maxpool_ties_and_lazy_input(cuda):
    """3x3 s2 p1 max-pool over relu(bn(y)): post-ReLU zeros tie; the FIRST max in scan order wins,
    as torch's CPU kernel does, so the argmax (and with it the backward) matches."""
    L = lib()
    g = torch.Generator().manual_seed(2)
    n, c, h, w = 2, 64, 16, 24
    y = torch.randn(n, c, h, w, generator=g)
    sc, sh = torch.rand(c, generator=g) + 0.5, torch.randn(c, generator=g) * 0.5 - 0.5
    a = torch.relu(y * sc[:, None, None] + sh[:, None, None])
    ref, ridx = F.max_pool2d(a, 3, 2, 1, return_indices=True)
    yd = nhwc(y).to(cuda)
    t = [sc.to(cuda), sh.to(cuda)]
    out = torch.empty(n, h // 2, w // 2, c, device=cuda)
    idx = torch.empty(n, h // 2, w // 2, c, dtype=torch.uint8, device=cuda)
    s0 = src(yd, t[0], t[1], relu=1)
    L.check(L.lib().uwm_op_maxpool(C.byref(s0), n, P(out), P(idx), stream()))
    torch.cuda.synchronize()
    got = nchw(out.cpu())
    # the GPU contracts y*s+b into one FMA (single rounding); torch CPU rounds twice -> <= 1 ulp apart
    assert (got - ref).abs().max() <= 4e-7 * max(1.0, float(ref.abs().max()))
    # decode our tap index to torch's flat input index; exact ties (post-ReLU zeros) must agree,
    # only last-ulp near-ties may pick a different tap
    tap = nchw(idx.cpu()).long()
    ho = torch.arange(h // 2).view(1, 1, -1, 1)
    wo = torch.arange(w // 2).view(1, 1, 1, -1)
    flat = (ho * 2 - 1 + tap // 3) * w + (wo * 2 - 1 + tap % 3)
    diff = flat != ridx
    assert diff.float().mean() < 1e-3
    if diff.any():
        af = a.flatten(2)
        v0 = af.gather(2, flat.flatten(2))[diff.flatten(2)]
        v1 = af.gather(2, ridx.flatten(2))[diff.flatten(2)]
        assert (v0 - v1).abs().max() <= 4e-7 * max(1.0, float(ref.abs().max()))
    assert torch.equal(flat[ref == 0], ridx[ref == 0])


@pytest.mark.parametrize("c,n,h,w", [(16, 2, 24, 40), (64, 3, 16, 16), (512, 4, 4, 5)])
def test_bn_backward_matches_autograd(cuda, c, n, h, w):
    """bn_bwd_reduce + bn_bwd_apply == autograd of F.batch_norm(training=True) in fp64."""
    L = lib()
    g = torch.Generator().manual_seed(11)
    y = (torch.randn(n, c, h, w, generator=g) * 2 + 0.7).double().requires_grad_()
    gamma = (torch.rand(c, generator=g) + 0.5).double().requires_grad_()
    beta = torch.randn(c, generator=g).double().requires_grad_()
    z = F.batch_norm(y, None, None, gamma, beta, True, 0.1, 1e-5)
    go = torch.randn(n, c, h, w, generator=g).double() + 0.5          # non-zero mean: exercises the cancellation
    z.backward(go)
    mean = y.detach().mean((0, 2, 3)); var = y.detach().var((0, 2, 3), unbiased=False)
    rstd = 1.0 / torch.sqrt(var + 1e-5)
    dev = lambda t: t.float().to(cuda)
    gd, yd = nhwc(go.float()).to(cuda), nhwc(y.detach().float()).to(cuda)
    dy = torch.empty_like(gd); dgam = torch.empty(c, device=cuda); dbet = torch.empty(c, device=cuda)
    scr = torch.empty(2 * c, dtype=torch.float64, device=cuda)
    t = [dev(mean), dev(rstd), dev(gamma.detach())]
    L.check(L.lib().uwm_op_bn_backward(P(gd), P(yd), P(t[0]), P(t[1]), P(t[2]), P(scr), P(dy), P(dgam), P(dbet), n * h * w, c,
                                       stream()))
    torch.cuda.synchronize()
    ref = y.grad.float()
    assert (nchw(dy.cpu()) - ref).abs().max() < 2e-5 * float(ref.abs().max()) + 1e-6
    assert torch.allclose(dgam.cpu(), gamma.grad.float(), rtol=1e-4, atol=1e-4 * float(gamma.grad.abs().max()))
    assert torch.allclose(dbet.cpu(), beta.grad.float(), rtol=1e-5, atol=1e-5 * float(beta.grad.abs().max()))


def test_upsplit_matches_autograd(cuda):
    L = lib()
    g = torch.Generator().manual_seed(12)
    n, c0, c1, h, w = 2, 32, 16, 8, 12
    prev = torch.randn(n, c0, h, w, generator=g, requires_grad=True)
    skip = torch.randn(n, c1, 2 * h, 2 * w, generator=g, requires_grad=True)
    sc, sh = torch.rand(c0, generator=g) + 0.5, torch.randn(c0, generator=g) * 0.3
    a = torch.relu(prev * sc[:, None, None] + sh[:, None, None])
    a.retain_grad()
    cat = torch.cat([F.interpolate(a, scale_factor=2, mode="nearest"), skip], 1)
    dcat = torch.randn(cat.shape, generator=g)
    cat.backward(dcat)
    # the kernel returns the gradient wrt the post-ReLU activation, masked by the ReLU: a.grad * (a > 0)
    ref_prev = (a.grad * (a.detach() > 0)).permute(0, 2, 3, 1)
    dd = nhwc(dcat).to(cuda); pm = nhwc(prev.detach()).to(cuda)
    t = [sc.to(cuda), sh.to(cuda)]
    gp = torch.empty(n, h, w, c0, device=cuda); gs = torch.empty(n, 2 * h, 2 * w, c1, device=cuda)
    L.check(L.lib().uwm_op_upsplit(P(dd), n, 2 * h, 2 * w, c0, c1, P(gp), P(pm), P(t[0]), P(t[1]), P(gs), stream()))
    torch.cuda.synchronize()
    assert (gp.cpu() - ref_prev).abs().max() < 1e-5
    assert torch.equal(gs.cpu(), skip.grad.permute(0, 2, 3, 1).contiguous())


@pytest.mark.parametrize("c0,c1,cout", [(32, 16, 16), (64, 64, 32), (32, 0, 16), (128, 64, 64)])      # (32, 0, 16): conv_up2_dgrad_kernel (sub-pixel form)
def test_dgrad_with_fused_concat_split(cuda, c0, c1, cout):
    """decoder conv1 backward: dgrad (Winograd) writes the 2x2-pooled, ReLU-masked gradient of up(prev) and the skip
    gradient straight from its epilogue — must equal autograd through cat(interpolate(relu(bn(prev))), skip) -> conv."""
    L = lib()
    g = torch.Generator().manual_seed(21)
    n, h, w = 2, 8, 16
    prev = torch.randn(n, c0, h, w, generator=g, requires_grad=True)
    sc, sh = torch.rand(c0, generator=g) + 0.5, torch.randn(c0, generator=g) * 0.3
    a = torch.relu(prev * sc[:, None, None] + sh[:, None, None])
    a.retain_grad()
    parts = [F.interpolate(a, scale_factor=2, mode="nearest")]
    skip = None
    if c1:
        skip = torch.randn(n, c1, 2 * h, 2 * w, generator=g, requires_grad=True)
        parts.append(skip)
    wt = (torch.randn(cout, c0 + c1, 3, 3, generator=g) * 0.05)
    y = F.conv2d(torch.cat(parts, 1), wt, None, 1, 1)
    dy = torch.randn(y.shape, generator=g)
    y.backward(dy)
    ref_prev = (a.grad * (a.detach() > 0)).permute(0, 2, 3, 1)
    wp, kpad = pack_w(wt)
    wp = wp.to(cuda)
    kpadd = rup(9 * cout, 32)
    wd = torch.empty(c0 + c1, kpadd, device=cuda)
    L.check(L.lib().uwm_op_pack_dgrad(P(wp), cout, kpad, 9, c0 + c1, P(wd), kpadd, cout, stream()))
    dyd = nhwc(dy).to(cuda); pm = nhwc(prev.detach()).to(cuda)
    t = [sc.to(cuda), sh.to(cuda)]
    gp = torch.full((n, h, w, c0), float("nan"), device=cuda)
    gs = torch.full((n, 2 * h, 2 * w, max(c1, 4)), float("nan"), device=cuda)
    L.check(L.lib().uwm_op_dgrad_upsplit(P(dyd), n, 2 * h, 2 * w, cout, P(wd), c0, c1, kpadd, P(gp), P(pm), P(t[0]), P(t[1]),
                                         P(gs) if c1 else None, stream()))
    torch.cuda.synchronize()
    assert (gp.cpu() - ref_prev).abs().max() < 3e-5 * max(1.0, float(ref_prev.abs().max()))
    if c1:
        ref_skip = skip.grad.permute(0, 2, 3, 1)
        assert (gs.cpu() - ref_skip).abs().max() < 3e-5 * max(1.0, float(ref_skip.abs().max()))


def test_residual_and_maxpool_backward(cuda):
    L = lib()
    g = torch.Generator().manual_seed(13)
    n, c, h, w = 2, 64, 16, 24
    # residual: relu(bn2(y2) + bn_d(yd))
    y2, yd = torch.randn(n, c, h, w, generator=g), torch.randn(n, c, h, w, generator=g)
    s2, b2, sd, bd = (torch.rand(c, generator=g) + 0.5, torch.randn(c, generator=g) * 0.2,
                      torch.rand(c, generator=g) + 0.5, torch.randn(c, generator=g) * 0.2)
    bc = lambda v: v[None, :, None, None]
    ref = torch.relu(y2 * bc(s2) + bc(b2) + yd * bc(sd) + bc(bd))
    out = torch.empty(n, h, w, c, device=cuda)
    t = [nhwc(y2).to(cuda), s2.to(cuda), b2.to(cuda), nhwc(yd).to(cuda), sd.to(cuda), bd.to(cuda)]
    L.check(L.lib().uwm_op_residual(P(t[0]), P(t[1]), P(t[2]), P(t[3]), P(t[4]), P(t[5]), P(out), n * h * w, c, stream()))
    torch.cuda.synchronize()
    assert (nchw(out.cpu()) - ref).abs().max() < 1e-5
    # max-pool backward through relu(bn(y)) with an added skip gradient
    y = torch.randn(n, c, h, w, generator=g)
    a = torch.relu(y * bc(s2) + bc(b2)).requires_grad_()
    pooled = F.max_pool2d(a, 3, 2, 1)
    go = torch.randn(pooled.shape, generator=g)
    pooled.backward(go)
    addend = torch.randn(n, h, w, c, generator=g)
    yd_ = nhwc(y).to(cuda); tt = [s2.to(cuda), b2.to(cuda)]
    s0 = src(yd_, tt[0], tt[1], relu=1)
    po = torch.empty(n, h // 2, w // 2, c, device=cuda); idx = torch.empty(n, h // 2, w // 2, c, dtype=torch.uint8, device=cuda)
    L.check(L.lib().uwm_op_maxpool(C.byref(s0), n, P(po), P(idx), stream()))
    gin = torch.empty(n, h, w, c, device=cuda)
    gd, ad = nhwc(go).to(cuda), addend.to(cuda)
    L.check(L.lib().uwm_op_maxpool_backward(P(gd), P(idx), P(ad), C.byref(s0), n, P(gin), stream()))
    torch.cuda.synchronize()
    refg = (a.grad.permute(0, 2, 3, 1) + addend) * (a.detach().permute(0, 2, 3, 1) > 0)
    # positive near-ties may route to another tap on one side only (FMA rounding): tolerate a handful
    bad = ((gin.cpu() - refg).abs() > 1e-5).float().mean()
    assert bad < 1e-3, bad


@pytest.mark.parametrize("shape", [
    # n, c, h, w, k, stride, pad_begin (efficientnet_pytorch static same padding: (0,1) | (1,1) | (2,2) | (1,2))
    (2, 48, 16, 24, 3, 1, 1),
    (2, 144, 32, 32, 3, 2, 0),       # pad (0,1)
    (1, 192, 24, 40, 5, 2, 2),       # pad (2,2), ragged bands (12 output rows, 20 columns)
    (3, 960, 8, 8, 5, 2, 1),         # pad (1,2), 240 channel quads
    (2, 336, 12, 20, 5, 1, 2),
    (1, 24, 7, 9, 3, 1, 1),          # odd sizes, 6 channel quads
    (1, 2688, 4, 4, 3, 1, 1),        # widest layer on a 4x4 map: almost every tap is padding
])
def test_depthwise_conv_forward_dgrad_wgrad(cuda, shape):
    """mbconv.hip depthwise kernels (tap-major weights, asymmetric static padding) against torch's grouped conv2d and its
    autograd: forward, dgrad (+ addend), wgrad (accumulating)."""
    L = lib()
    n, c, h, w, k, s, pb = shape
    g = torch.Generator().manual_seed(11)
    ho, wo = -(-h // s), -(-w // s)
    pe_h = max((ho - 1) * s + k - h - pb, 0); pe_w = max((wo - 1) * s + k - w - pb, 0)
    x = torch.randn(n, c, h, w, generator=g, requires_grad=True)
    wt = (torch.randn(c, 1, k, k, generator=g) * 0.2).requires_grad_()
    y = F.conv2d(F.pad(x, (pb, pe_w, pb, pe_h)), wt, None, s, 0, groups=c)
    assert y.shape[-2:] == (ho, wo)
    dy = torch.randn(y.shape, generator=g)
    y.backward(dy)
    xd, dyd = nhwc(x.detach()).to(cuda), nhwc(dy).to(cuda)
    wd = wt.detach().reshape(c, k * k).t().contiguous().to(cuda)           # [k*k][C]
    out = torch.full((n, ho, wo, c), float("nan"), device=cuda)
    L.check(L.lib().uwm_op_depthwise(0, P(xd), P(wd), k, s, pb, n, h, w, c, ho, wo, None, P(out), None, stream()))
    add = torch.randn(n, h, w, c, generator=g)
    dx = torch.full((n, h, w, c), float("nan"), device=cuda)
    L.check(L.lib().uwm_op_depthwise(1, P(dyd), P(wd), k, s, pb, n, h, w, c, ho, wo, P(add.to(cuda)), P(dx), None, stream()))
    dw0 = torch.randn(k * k, c, generator=g)
    dw = dw0.to(cuda)
    scr = torch.empty(max(1, L.lib().uwm_op_depthwise_scratch_floats(k, n, c, ho, wo)), device=cuda)
    L.check(L.lib().uwm_op_depthwise(2, P(xd), P(dyd), k, s, pb, n, h, w, c, ho, wo, None, P(dw), P(scr), stream()))
    torch.cuda.synchronize()
    assert (nchw(out.cpu(), c) - y.detach()).abs().max() < 2e-5 * max(1.0, float(y.abs().max()))
    ref_dx = x.grad.permute(0, 2, 3, 1) + add
    assert (dx.cpu() - ref_dx).abs().max() < 2e-5 * max(1.0, float(ref_dx.abs().max()))
    ref_dw = wt.grad.reshape(c, k * k).t() + dw0
    assert (dw.cpu() - ref_dw).abs().max() < 5e-5 * max(1.0, float(ref_dw.abs().max()))


@pytest.mark.parametrize("shape", [
    (2, 64, 64, 24, 40),       # layer1-like, partial 16x16 tiles in both directions
    (1, 32, 128, 16, 16),      # one chunk pair, two channel tiles
    (2, 128, 96, 8, 16),       # minimum tile height; Cout 96 over-hangs the 64-channel tile
    (1, 256, 256, 32, 32),     # deep stage
])
def test_conv_f16x3_direct(cuda, shape):
    """conv_f16x3.hip (cfg 600): direct 3x3 on v_mfma_f32_16x16x32_f16 with every operand split into two fp16 halves and
    row-scaled filters — the fp32 bars of _conv_case (2e-5 of the output range, BatchNorm statistics 1e-5), plain and with the
    lazy BatchNorm + ReLU input transform."""
    n, cin, cout, h, w = shape
    _conv_case(cuda, n, cin, cout, h, w, 3, 1, 1, cfg=600)
    _conv_case(cuda, n, cin, cout, h, w, 3, 1, 1, cfg=600, lazy=True, seed=5)
    # the launcher picks the kernel by launch size (these shapes are small: the eight-wave kernel); force the other ones too:
    # 601 = four waves, 64-channel tiles; 603 = four waves, 32-channel tiles (the 32-output layers)
    _conv_case(cuda, n, cin, cout, h, w, 3, 1, 1, cfg=601, lazy=True, seed=6)
    _conv_case(cuda, n, cin, cout, h, w, 3, 1, 1, cfg=602, seed=7)
    _conv_case(cuda, n, cin, cout, h, w, 3, 1, 1, cfg=603, lazy=True, seed=8)
    _conv_case(cuda, n, cin, 32, h, w, 3, 1, 1, cfg=603, seed=9)


def test_conv_stem_f16x3(cuda):
    """conv_stem_f16x3.hip (cfg 610): the 7x7 / stride-2 / pad-3 stem (3 input channels stored as 4) with one MFMA k-step per kernel
    row, fp16x3 products: the fp32 bars of _conv_case (output 2e-5 of its range, BatchNorm statistics), whole and partial tiles,
    odd image sizes."""
    for n, h, w in ((2, 64, 64), (1, 96, 80), (3, 38, 50), (1, 33, 47)):
        _conv_case(cuda, n, 3, 64, h, w, 7, 2, 3, cfg=610, seed=h)


def test_conv_f16x3_single_chunk_16_to_16(cuda):
    """the single-chunk form of conv_f16x3 (16 -> 16 channels: decoder block 4 conv2), plain and with a lazy BatchNorm + ReLU input,
    whole and partial 16x16 tiles."""
    for n, h, w in ((2, 32, 48), (1, 24, 40), (3, 16, 16)):
        _conv_case(cuda, n, 16, 16, h, w, 3, 1, 1, cfg=600, seed=h)
        _conv_case(cuda, n, 16, 16, h, w, 3, 1, 1, cfg=600, lazy=True, seed=h + 1)


def test_conv_f16x3_error_vs_fp64_and_range(cuda):
    """fp16x3 keeps 22 mantissa bits per operand: against an fp64 convolution its error must stay within 4x the exact-fp32
    direct kernel's (the bf16x3 kernel's is ~30x) — and that must hold when the operands sit far from fp16's comfortable range:
    tiny filters (1e-6: scaled up row by row), rows of very different magnitude, large activations.  Activations are NOT
    scaled (a power of two that is safe for every input does not exist without a pass over the tensor): below the fp16
    normal range the low half is a subnormal with an absolute step of 2^-24, so a tensor whose values are all tiny keeps an
    absolute error of ~3e-8 per element — 3e-8 / |x| relative (last case; unit-scale activations behind a BatchNorm: the fp32 level)."""
    L = lib()
    g = torch.Generator().manual_seed(11)
    n, cin, cout, h, w = 2, 128, 64, 32, 32
    for xmag, wmag in ((1.0, 1.0), (300.0, 1e-6), (0.25, 3.0), (1e-3, 20.0)):
        x = torch.randn(n, cin, h, w, generator=g) * xmag
        wt = torch.randn(cout, cin, 3, 3, generator=g) * (wmag / (cin * 9) ** 0.5)
        wt[5] *= 1e-4; wt[7] *= 1e3                      # rows of very different magnitude: each has its own scale
        ref = F.conv2d(x.double(), wt.double(), None, 1, 1)
        xd = nhwc(x).to(cuda)
        wp, kpad = pack_w(wt)
        wp = wp.to(cuda)
        errs = {}
        for name, cfg in (("f16x3", 600), ("direct", 164)):
            y = torch.empty(n, h, w, cout, device=cuda)
            s0 = src(xd)
            L.check(L.lib().uwm_op_conv(C.byref(s0), None, P(wp), cout, kpad, 3, 3, 1, 1, n, cout, None, P(y), None, cfg, stream()))
            torch.cuda.synchronize()
            d = (nchw(y.cpu()).double() - ref).abs()
            errs[name] = float((d / ref.abs().amax((0, 2, 3), keepdim=True)).max())       # relative to each output channel's range
        bound = 4 * errs["direct"] + 1e-7 if xmag >= 0.25 else 4e-8 / xmag
        assert errs["f16x3"] < bound and errs["f16x3"] < 5e-5, (xmag, wmag, errs)


@pytest.mark.parametrize("shape", [
    (2, 64, 128, 32, 32, 3, 2, 1),    # layer2.0.conv1: 3x3 stride 2
    (2, 64, 128, 32, 32, 1, 2, 0),    # downsample 1x1 stride 2
    (1, 128, 256, 16, 24, 3, 2, 1),   # layer3.0.conv1, ragged pixel tiles
    (3, 256, 512, 8, 8, 1, 2, 0),     # layer4 downsample
])
def test_igemm_f16x3_stride2_layers(cuda, shape):
    """The stride-2 layers of the ResNet encoders (3x3 / stride 2, 1x1 / stride 2) and their dgrads on the implicit GEMM with
    fp16x3 split products (ConvArgs::ig16 — conv_igemm_kernel<..., F16 = true>; the model takes it for these layers in the fp16x3
    precision modes): forward with a lazy BatchNorm + ReLU source and statistics, dgrad with addend + ReLU mask and a dY as tiny
    as a real Dice gradient (scaled through max|dY|), held against fp64 convolutions within 4x the exact-fp32 kernel's own error."""
    L = lib()
    n, cin, cout, h, w, k, st, p = shape
    g = torch.Generator().manual_seed(9)
    x = torch.randn(n, cin, h, w, generator=g)
    wt = torch.randn(cout, cin, k, k, generator=g) * (2.0 / (cin * k * k)) ** 0.5
    sc = torch.rand(cin, generator=g) + 0.5; sc[::5] *= -1
    sh = torch.randn(cin, generator=g) * 0.2
    act = torch.relu(x * sc[:, None, None] + sh[:, None, None])
    ref = F.conv2d(act.double(), wt.double(), None, st, p)
    ho, wo = ref.shape[-2:]
    xd = nhwc(x).to(cuda); scd, shd = sc.to(cuda), sh.to(cuda)
    wp, kpad = pack_w(wt)
    wp = wp.to(cuda)
    errs = {}
    for name, on in (("f32", 0), ("f16x3", 1)):
        L.lib().uwm_op_set_igemm_f16x3(on)
        try:
            y = torch.empty(n, ho, wo, cout, device=cuda)
            stats = torch.zeros(2 * cout, dtype=torch.float64, device=cuda)
            s0 = src(xd, scd, shd, relu=1)
            L.check(L.lib().uwm_op_conv(C.byref(s0), None, P(wp), cout, kpad, k, k, st, p, n, cout, None, P(y), P(stats), -1, stream()))
            torch.cuda.synchronize()
        finally:
            L.lib().uwm_op_set_igemm_f16x3(0)
        d = (nchw(y.cpu()).double() - ref).abs()
        errs[name] = float((d / ref.abs().amax((0, 2, 3), keepdim=True)).max())
        assert torch.allclose(stats[:cout].cpu(), ref.sum((0, 2, 3)), rtol=1e-4, atol=1e-3 * float(ref.abs().max()))
    assert errs["f16x3"] < 4 * errs["f32"] + 1e-7 and errs["f16x3"] < 2e-5, errs
    # ---- dgrad (transposed gather by output-pixel parity classes), dY ~ 1e-6
    xg = torch.randn(n, cin, h, w, generator=g).double().requires_grad_()
    yg = F.conv2d(xg, wt.double(), None, st, p)
    dy = torch.randn(yg.shape, generator=g) * 1e-6
    yg.backward(dy.double())
    coutp = rup(cout, 4)
    kpadd = rup(k * k * coutp, 32)
    wd = torch.empty(cin, kpadd, device=cuda)
    L.check(L.lib().uwm_op_pack_dgrad(P(wp), cout, kpad, k * k, cin, P(wd), kpadd, coutp, stream()))
    dyd = nhwc(dy).to(cuda)
    addend = torch.randn(n, h, w, cin, generator=g) * 1e-6
    maskt = torch.randn(n, h, w, cin, generator=g)
    add_d, mask_d = addend.to(cuda), maskt.to(cuda)
    ref_dx = (xg.grad.permute(0, 2, 3, 1) + addend.double()) * (maskt > 0)
    derr = {}
    for name, on in (("f32", 0), ("f16x3", 1)):
        L.lib().uwm_op_set_igemm_f16x3(on)
        try:
            dx = torch.empty(n, h, w, cin, device=cuda)
            L.check(L.lib().uwm_op_dgrad(P(dyd), n, ho, wo, coutp, P(wd), cin, kpadd, k, k, st, p, h, w, P(add_d), P(mask_d), None, None,
                                         P(dx), stream()))
            torch.cuda.synchronize()
        finally:
            L.lib().uwm_op_set_igemm_f16x3(0)
        derr[name] = float((dx.cpu().double() - ref_dx).abs().max() / ref_dx.abs().max())
    assert derr["f16x3"] < 4 * derr["f32"] + 1e-7 and derr["f16x3"] < 2e-5, derr


@pytest.mark.parametrize("n,hs,ws,lazy", [(2, 16, 32, True), (1, 8, 16, False), (3, 24, 48, True)])
def test_conv_up2_f16x3_forward_and_dgrad(cuda, n, hs, ws, lazy):
    """decoder block 4 conv1 (sub-pixel form) on conv_up2_f16.hip — the fp16x3 split products on v_mfma_f32_16x16x32_f16 the model
    takes in the fp16x3 precision modes (ConvArgs::ig16): forward output + BatchNorm statistics and the dgrad with the fused concat
    split (2x2-pooled, ReLU-masked gradient of the low-resolution producer; dY as tiny as a real Dice gradient, scaled through
    max|dY|) against fp64, within 4x the exact-fp32 sub-pixel kernel's own error."""
    L = lib()
    g = torch.Generator().manual_seed(41 + hs)
    a = torch.randn(n, 32, hs, ws, generator=g) * 3.0
    sc = torch.rand(32, generator=g) + 0.5; sh = torch.randn(32, generator=g) * 0.3
    sc[::7] *= -1
    wt = torch.randn(16, 32, 3, 3, generator=g) * 0.08
    act = torch.relu(a * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1)) if lazy else a
    y = F.conv2d(F.interpolate(act.double(), scale_factor=2, mode="nearest"), wt.double(), None, padding=1)
    ad = nhwc(a.to(cuda)); wp, kpad = pack_w(wt.to(cuda))
    t = [sc.to(cuda), sh.to(cuda)]
    s0 = src(ad, t[0], t[1], relu=1, up=1) if lazy else src(ad, up=1)
    errs = {}
    for name, on in (("f32", 0), ("f16x3", 1)):
        L.lib().uwm_op_set_igemm_f16x3(on)
        try:
            out = torch.full((n, 2 * hs, 2 * ws, 16), float("nan"), device=cuda)
            stats = torch.zeros(32, dtype=torch.float64, device=cuda)
            L.check(L.lib().uwm_op_conv(C.byref(s0), None, P(wp), 16, kpad, 3, 3, 1, 1, n, 16, None, P(out), P(stats), -1, stream()))
            torch.cuda.synchronize()
        finally:
            L.lib().uwm_op_set_igemm_f16x3(0)
        got = nchw(out.cpu(), 16).double()
        errs[name] = float((got - y).abs().max() / y.abs().max())
        assert torch.allclose(stats[:16].cpu(), y.sum((0, 2, 3)), rtol=1e-5, atol=1e-3)
        assert torch.allclose(stats[16:].cpu(), (y * y).sum((0, 2, 3)), rtol=1e-5, atol=1e-3)
    assert errs["f16x3"] < 4 * errs["f32"] + 1e-7 and errs["f16x3"] < 1e-5, errs
    # ---- dgrad with the fused concat split
    prev = (torch.randn(n, 32, hs, ws, generator=g)).double().requires_grad_()
    ap = torch.relu(prev * sc.double()[:, None, None] + sh.double()[:, None, None])
    ap.retain_grad()
    yy = F.conv2d(F.interpolate(ap, scale_factor=2, mode="nearest"), wt.double(), None, 1, 1)
    dy = torch.randn(yy.shape, generator=g) * 1e-6
    yy.backward(dy.double())
    ref_prev = (ap.grad * (ap.detach() > 0)).permute(0, 2, 3, 1)
    kpadd = rup(9 * 16, 32)
    wd = torch.empty(32, kpadd, device=cuda)
    L.check(L.lib().uwm_op_pack_dgrad(P(wp), 16, kpad, 9, 32, P(wd), kpadd, 16, stream()))
    dyd = nhwc(dy).to(cuda); pm = nhwc(prev.detach().float()).to(cuda)
    derr = {}
    for name, on in (("f32", 0), ("f16x3", 1)):
        L.lib().uwm_op_set_igemm_f16x3(on)
        try:
            gp = torch.full((n, hs, ws, 32), float("nan"), device=cuda)
            L.check(L.lib().uwm_op_dgrad_upsplit(P(dyd), n, 2 * hs, 2 * ws, 16, P(wd), 32, 0, kpadd, P(gp), P(pm), P(t[0]), P(t[1]), None, stream()))
            torch.cuda.synchronize()
        finally:
            L.lib().uwm_op_set_igemm_f16x3(0)
        derr[name] = float((gp.cpu().double() - ref_prev).abs().max() / ref_prev.abs().max())
    assert derr["f16x3"] < 4 * derr["f32"] + 1e-7 and derr["f16x3"] < 1e-5, derr


@pytest.mark.parametrize("n,h,w,ch", [(2, 16, 64, 16), (1, 8, 32, 16), (3, 40, 96, 16), (2, 16, 64, 32), (1, 8, 32, 32), (3, 24, 96, 32), (12, 64, 256, 32), (10, 64, 256, 16)])      # (the last two: several tiles per persistent workgroup)
def test_conv_c16_f16x3_forward_and_dgrad(cuda, n, h, w, ch):
    """decoder block 4 conv2 (3x3, 16 -> 16 at full resolution) on conv_c16_f16.hip (ConvArgs::ig16; cfg 710 forced, -1 auto) and block 3 conv2 (32 -> 32, conv_c32_f16_kernel: cfg 711): tap
    pairs as MFMA k-steps, persistent double-buffered patch.  Forward: lazy BatchNorm + ReLU source (negative scales), output and
    BatchNorm statistics; dgrad: dY ~ 1e-6 (scaled through max|dY|), addend + ReLU mask with lazy mask scale / shift.  Against fp64
    within 4x the exact-fp32 routes' own error."""
    L = lib()
    g = torch.Generator().manual_seed(81 + h)
    x = torch.randn(n, ch, h, w, generator=g) * 2.0
    sc = torch.rand(ch, generator=g) + 0.5; sc[::5] *= -1
    sh = torch.randn(ch, generator=g) * 0.3
    wt = torch.randn(ch, ch, 3, 3, generator=g) * 0.1
    act = torch.relu(x * sc[:, None, None] + sh[:, None, None])
    ref = F.conv2d(act.double(), wt.double(), None, 1, 1)
    xd = nhwc(x).to(cuda); scd, shd = sc.to(cuda), sh.to(cuda)
    wp, kpad = pack_w(wt)
    wp = wp.to(cuda)
    errs = {}
    for name, on, cfg in (("f32", 0, -1), ("f16x3", 1, -1), ("forced", 1, 710 if ch == 16 else 711)):
        L.lib().uwm_op_set_igemm_f16x3(on)
        try:
            y = torch.full((n, h, w, ch), float("nan"), device=cuda)
            stats = torch.zeros(2 * ch, dtype=torch.float64, device=cuda)
            s0 = src(xd, scd, shd, relu=1)
            L.check(L.lib().uwm_op_conv(C.byref(s0), None, P(wp), ch, kpad, 3, 3, 1, 1, n, ch, None, P(y), P(stats), cfg, stream()))
            torch.cuda.synchronize()
        finally:
            L.lib().uwm_op_set_igemm_f16x3(0)
        errs[name] = float((nchw(y.cpu()).double() - ref).abs().max() / ref.abs().max())
        assert torch.allclose(stats[:ch].cpu(), ref.sum((0, 2, 3)), rtol=1e-5, atol=1e-3 * float(ref.abs().max()))
        assert torch.allclose(stats[ch:].cpu(), (ref * ref).sum((0, 2, 3)), rtol=1e-5, atol=1e-3 * float((ref * ref).sum((0, 2, 3)).max()))
        if name == "f16x3": y16 = y.clone()
        if name == "forced": assert torch.equal(y, y16)                     # (the auto route under ig16 IS this kernel)
    assert errs["f16x3"] < 4 * errs["f32"] + 1e-7 and errs["f16x3"] < 1e-5, errs
    assert errs["f16x3"] != errs["f32"]
    # ---- dgrad with addend + lazily activated ReLU mask
    xg = torch.randn(n, ch, h, w, generator=g).double().requires_grad_()
    yg = F.conv2d(xg, wt.double(), None, 1, 1)
    dy = torch.randn(yg.shape, generator=g) * 1e-6
    yg.backward(dy.double())
    kpadd = rup(9 * ch, 32)
    wd = torch.empty(ch, kpadd, device=cuda)
    L.check(L.lib().uwm_op_pack_dgrad(P(wp), ch, kpad, 9, ch, P(wd), kpadd, ch, stream()))
    dyd = nhwc(dy).to(cuda)
    addend = torch.randn(n, h, w, ch, generator=g) * 1e-6
    maskt = torch.randn(n, h, w, ch, generator=g)
    add_d, mask_d = addend.to(cuda), maskt.to(cuda)
    live = (maskt * sc + sh) > 0
    ref_dx = (xg.grad.permute(0, 2, 3, 1) + addend.double()) * live
    derr = {}
    for name, on in (("f32", 0), ("f16x3", 1)):
        L.lib().uwm_op_set_igemm_f16x3(on)
        try:
            dx = torch.full((n, h, w, ch), float("nan"), device=cuda)
            L.check(L.lib().uwm_op_dgrad(P(dyd), n, h, w, ch, P(wd), ch, kpadd, 3, 3, 1, 1, h, w, P(add_d), P(mask_d), P(scd), P(shd), P(dx), stream()))
            torch.cuda.synchronize()
        finally:
            L.lib().uwm_op_set_igemm_f16x3(0)
        derr[name] = float((dx.cpu().double() - ref_dx).abs().max() / ref_dx.abs().max())
    assert derr["f16x3"] < 4 * derr["f32"] + 1e-7 and derr["f16x3"] < 1e-5, derr
    assert derr["f16x3"] != derr["f32"]


@pytest.mark.parametrize("n,cin,cout,h,w,cfg", [(2, 64, 256, 32, 48, 864), (1, 256, 64, 64, 64, 864), (1, 512, 128, 32, 32, 928), (4, 256, 64, 128, 128, -1)])
def test_conv_gemm_f16x3_1x1(cuda, n, cin, cout, h, w, cfg):
    """conv_gemm.hip's F16 form (ConvArgs::ig16; the Bottleneck 1x1 / stride-1 layers in the fp16x3 precision modes): raw fp32 chunks by
    LDS-DMA, operands split into fp16 hi / lo halves after ds_read, one v_mfma_f32_16x16x32_f16 k-step per 32-channel chunk.  Forward
    with a lazy BatchNorm + ReLU source (negative scales) and statistics (cfg 864 / 928 = the 64- / 128-channel tiles forced, -1 = auto
    on a launch big enough for the persistent GEMM); dgrad (auto route) with dY ~ 1e-6, addend and ReLU mask.  Against fp64 within 4x
    the exact-fp32 kernel's own error."""
    L = lib()
    g = torch.Generator().manual_seed(91 + cin)
    x = torch.randn(n, cin, h, w, generator=g) * 2.0
    wt = torch.randn(cout, cin, 1, 1, generator=g) * (2.0 / cin) ** 0.5
    sc = torch.rand(cin, generator=g) + 0.5; sc[::5] *= -1
    sh = torch.randn(cin, generator=g) * 0.2
    act = torch.relu(x * sc[:, None, None] + sh[:, None, None])
    ref = F.conv2d(act.double(), wt.double())
    xd = nhwc(x).to(cuda); scd, shd = sc.to(cuda), sh.to(cuda)
    wp, kpad = pack_w(wt)
    wp = wp.to(cuda)
    errs = {}
    for name, on in (("f32", 0), ("f16x3", 1)):
        L.lib().uwm_op_set_igemm_f16x3(on)
        try:
            y = torch.full((n, h, w, cout), float("nan"), device=cuda)
            stats = torch.zeros(2 * cout, dtype=torch.float64, device=cuda)
            s0 = src(xd, scd, shd, relu=1)
            L.check(L.lib().uwm_op_conv(C.byref(s0), None, P(wp), cout, kpad, 1, 1, 1, 0, n, cout, None, P(y), P(stats), cfg, stream()))
            torch.cuda.synchronize()
        finally:
            L.lib().uwm_op_set_igemm_f16x3(0)
        d = (nchw(y.cpu()).double() - ref).abs()
        errs[name] = float((d / ref.abs().amax((0, 2, 3), keepdim=True)).max())
        assert torch.allclose(stats[:cout].cpu(), ref.sum((0, 2, 3)), rtol=1e-4, atol=1e-3 * float(ref.abs().max()))
    assert errs["f16x3"] < 4 * errs["f32"] + 1e-7 and errs["f16x3"] < 2e-5, errs
    assert errs["f16x3"] != errs["f32"]
    # ---- dgrad: K = the layer's OUTPUT channels
    xg = torch.randn(n, cin, h, w, generator=g).double().requires_grad_()
    yg = F.conv2d(xg, wt.double())
    dy = torch.randn(yg.shape, generator=g) * 1e-6
    yg.backward(dy.double())
    kpadd = rup(cout, 32)
    wd = torch.empty(cin, kpadd, device=cuda)
    L.check(L.lib().uwm_op_pack_dgrad(P(wp), cout, kpad, 1, cin, P(wd), kpadd, cout, stream()))
    dyd = nhwc(dy).to(cuda)
    addend = torch.randn(n, h, w, cin, generator=g) * 1e-6
    maskt = torch.randn(n, h, w, cin, generator=g)
    add_d, mask_d = addend.to(cuda), maskt.to(cuda)
    ref_dx = (xg.grad.permute(0, 2, 3, 1) + addend.double()) * (maskt > 0)
    derr = {}
    for name, on in (("f32", 0), ("f16x3", 1)):
        L.lib().uwm_op_set_igemm_f16x3(on)
        try:
            dx = torch.full((n, h, w, cin), float("nan"), device=cuda)
            L.check(L.lib().uwm_op_dgrad(P(dyd), n, h, w, cout, P(wd), cin, kpadd, 1, 1, 1, 0, h, w, P(add_d), P(mask_d), None, None, P(dx), stream()))
            torch.cuda.synchronize()
        finally:
            L.lib().uwm_op_set_igemm_f16x3(0)
        derr[name] = float((dx.cpu().double() - ref_dx).abs().max() / ref_dx.abs().max())
    assert derr["f16x3"] < 4 * derr["f32"] + 1e-7 and derr["f16x3"] < 2e-5, derr
    assert derr["f16x3"] != derr["f32"]


def test_conv_f16x3_upsample_concat(cuda):
    """decoder conv1 on the fp16x3 kernel: cat(nearest_x2(d), skip), both sources lazy, concat boundary on a 16-channel chunk."""
    L = lib()
    g = torch.Generator().manual_seed(1)
    n, c0, c1, cout, h, w = 2, 48, 16, 64, 8, 12
    d = torch.randn(n, c0, h, w, generator=g)
    sk = torch.randn(n, c1, 2 * h, 2 * w, generator=g)
    sc0, sh0 = torch.rand(c0, generator=g) + 0.5, torch.randn(c0, generator=g) * 0.2
    sc1, sh1 = torch.rand(c1, generator=g) + 0.5, torch.randn(c1, generator=g) * 0.2
    wt = torch.randn(cout, c0 + c1, 3, 3, generator=g) * 0.05
    a0 = torch.relu(d * sc0[:, None, None] + sh0[:, None, None])
    a1 = torch.relu(sk * sc1[:, None, None] + sh1[:, None, None])
    ref = F.conv2d(torch.cat([F.interpolate(a0, scale_factor=2, mode="nearest"), a1], 1), wt, None, 1, 1)
    dd, skd = nhwc(d).to(cuda), nhwc(sk).to(cuda)
    wp, kpad = pack_w(wt)
    wp = wp.to(cuda)
    t = [sc0.to(cuda), sh0.to(cuda), sc1.to(cuda), sh1.to(cuda)]
    y = torch.empty(n, 2 * h, 2 * w, cout, device=cuda)
    s0, s1 = src(dd, t[0], t[1], relu=1, up=1), src(skd, t[2], t[3], relu=1)
    L.check(L.lib().uwm_op_conv(C.byref(s0), C.byref(s1), P(wp), cout, kpad, 3, 3, 1, 1, n, cout, None, P(y), None, 600, stream()))
    torch.cuda.synchronize()
    assert (nchw(y.cpu()) - ref).abs().max() < 2e-5 * float(ref.abs().max())


@pytest.mark.parametrize("case", [
    (2, 64, 64, 16, 32, 1.0),        # layer1-like: one output-channel tile, two input-channel tiles
    (1, 32, 128, 8, 64, 1e-6),       # two output-channel tiles; dY as tiny as a real Dice gradient (scaled through xmax)
    (3, 96, 80, 12, 32, 1e-3),       # Cout 80 over-hangs the 64-channel tile; odd stage counts per split
    (4, 64, 128, 16, 16, 1e-5),      # 16-pixel-wide map (layer4-like): 8 x 16-pixel stages, two image rows per k-step
    (2, 32, 64, 8, 48, 1e-2),        # width 48: not a multiple of 32, three 16-pixel stage columns
])
def test_wgrad_f16x3_direct(cuda, case):
    """wgrad_f16x3.hip (force 6): direct 3x3 weight gradient on v_mfma_f32_16x16x32_f16, pixels as the MFMA K dimension
    (transposing LDS reads), fp16x3 split products with dY scaled by the power of two of its maximum: the fp32 bar of the
    other weight-gradient kernels (3e-5 of the gradient's range), with a lazy BatchNorm + ReLU input."""
    L = lib()
    n, cin, cout, h, w, dmag = case
    g = torch.Generator().manual_seed(13)
    x = torch.randn(n, cin, h, w, generator=g)
    sc, sh = torch.rand(cin, generator=g) + 0.5, torch.randn(cin, generator=g) * 0.2
    xin = torch.relu(x * sc[:, None, None] + sh[:, None, None])
    wt = (torch.randn(cout, cin, 3, 3, generator=g) * 0.05).requires_grad_()
    y = F.conv2d(xin.double(), wt.double(), None, 1, 1)
    dy = torch.randn(y.shape, generator=g) * dmag
    dy[:, :, :, : w // 2] *= 1e-3                         # a wide dynamic range inside the tensor
    ref = torch.autograd.grad(y, wt, dy.double())[0]
    xd, dyd = nhwc(x).to(cuda), nhwc(dy).to(cuda)
    kpad = rup(9 * cin, 32)
    dw = torch.zeros(cout, kpad, device=cuda)
    scd, shd = sc.to(cuda), sh.to(cuda)                    # (src() keeps nothing alive)
    s0 = src(xd, scd, shd, relu=1)
    L.check(L.lib().uwm_op_wgrad(C.byref(s0), None, P(dyd), n, h, w, cout, cout, kpad, 3, 3, 1, 1, P(dw), 6, stream()))
    torch.cuda.synchronize()
    got = unpack_w(dw.cpu(), cout, cin, 3, 3).double()
    assert torch.isfinite(got).all()
    assert (got - ref).abs().max() < 3e-5 * float(ref.abs().max()), float((got - ref).abs().max() / ref.abs().max())
    # twice the same launch: bit-identical (partial images + ordered reduce, no atomics)
    dw2 = torch.zeros(cout, kpad, device=cuda)
    L.check(L.lib().uwm_op_wgrad(C.byref(s0), None, P(dyd), n, h, w, cout, cout, kpad, 3, 3, 1, 1, P(dw2), 6, stream()))
    torch.cuda.synchronize()
    assert torch.equal(dw, dw2)


def test_wgrad_f16x3_upsample_concat(cuda):
    """decoder conv1's weight gradient on the fp16x3 kernel: cat(nearest_x2(d), skip) with the concat boundary on a 32-channel tile."""
    L = lib()
    g = torch.Generator().manual_seed(7)
    n, c0, c1, cout, h, w = 2, 64, 32, 32, 8, 16
    d = torch.randn(n, c0, h, w, generator=g)
    sk = torch.randn(n, c1, 2 * h, 2 * w, generator=g)
    sc0, sh0 = torch.rand(c0, generator=g) + 0.5, torch.randn(c0, generator=g) * 0.2
    a0 = torch.relu(d * sc0[:, None, None] + sh0[:, None, None])
    xin = torch.cat([F.interpolate(a0, scale_factor=2, mode="nearest"), sk], 1)
    wt = (torch.randn(cout, c0 + c1, 3, 3, generator=g) * 0.05).requires_grad_()
    y = F.conv2d(xin, wt, None, 1, 1)
    dy = torch.randn(y.shape, generator=g) * 1e-4
    y.backward(dy)
    dd, skd, dyd = nhwc(d).to(cuda), nhwc(sk).to(cuda), nhwc(dy).to(cuda)
    kpad = rup(9 * (c0 + c1), 32)
    dw = torch.zeros(cout, kpad, device=cuda)
    sc0d, sh0d = sc0.to(cuda), sh0.to(cuda)
    s0, s1 = src(dd, sc0d, sh0d, relu=1, up=1), src(skd)
    L.check(L.lib().uwm_op_wgrad(C.byref(s0), C.byref(s1), P(dyd), n, 2 * h, 2 * w, cout, cout, kpad, 3, 3, 1, 1, P(dw), 6, stream()))
    torch.cuda.synchronize()
    got = unpack_w(dw.cpu(), cout, c0 + c1, 3, 3)
    assert (got - wt.grad).abs().max() < 3e-5 * float(wt.grad.abs().max())
