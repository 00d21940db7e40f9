"""CPU tests of the host layer: the C-ABI library loads and exports every symbol include/uwm.h declares,
the parameter arena / state_dict contract, constructor validation, and loud failure without a GPU.
No compute call is made here (there is no GPU in this container)."""
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def U():
    import __graft_entry__ as g
    g.build()
    import unet_watermark_amd as U
    return U


def test_abi_exports_every_declared_symbol(U):
    from unet_watermark_amd import _lib as L
    hdr = open(os.path.join(ROOT, "include", "uwm.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)          # declarations only, not prose
    declared = set(re.findall(r"\b(uwm_[a-z0-9_]+)\s*\(", hdr))
    declared -= {"uwm_model"}
    lib = L.lib()
    for name in sorted(declared):
        assert hasattr(lib, name), f"{name} declared in include/uwm.h but not exported by libuwm.so"
    assert declared == set(L.SIGNATURES), declared ^ set(L.SIGNATURES)
    assert lib.uwm_version() >= 1


def test_state_dict_contract_matches_oracle(U):
    from oracle import unet_oracle as O
    for enc, count in (("resnet18", 14_328_209), ("resnet34", 24_436_369)):
        m, ref = U.Unet(enc), O.build(enc)
        sd, so = m.state_dict(), ref.state_dict()
        assert list(sd.keys()) == list(so.keys())
        assert all(sd[k].shape == so[k].shape and sd[k].dtype == so[k].dtype for k in so)
        assert m.num_parameters() == count == sum(p.numel() for p in m.parameters())
        m.load_state_dict(so)
        assert all(torch.equal(m.state_dict()[k], so[k]) for k in so)
        # padding of the arena stays exactly zero: sum over the arena == sum over the logical tensors
        logical = sum(float(v.double().abs().sum()) for k, v in so.items() if v.dtype.is_floating_point and "running" not in k)
        assert abs(float(m.flat_parameters().double().abs().sum()) - logical) < 1e-6 * logical
        # stage (gradient bucket) ranges tile the arena
        assert m.stages[0][0] == 0 and m.stages[-1][1] == m.flat_parameters().numel()
        assert all(m.stages[i][1] == m.stages[i + 1][0] for i in range(len(m.stages) - 1))


def test_efficientnet_b4_state_dict_contract(U):
    """EfficientNet-b4 encoder (BASELINE config 4) under both decoders: smp state_dict keys / shapes, parameter count,
    a loss-free load_state_dict round trip, BatchNorm hyper-parameters of the encoder, torch-default conv init."""
    from oracle import unet_oracle as O
    for arch, count in (("Unet", 19_419_289), ("UnetPlusPlus", 20_006_713)):
        torch.manual_seed(1)
        m, ref = getattr(U, arch)("efficientnet-b4"), O.build("efficientnet-b4", arch=arch)
        sd, so = m.state_dict(), ref.state_dict()
        assert list(sd.keys()) == list(so.keys())
        assert all(sd[k].shape == so[k].shape and sd[k].dtype == so[k].dtype for k in so)
        assert m.num_parameters() == count == sum(p.numel() for p in m.parameters())
        w = sd["encoder._blocks.2._expand_conv.weight"]        # kaiming_uniform(a=sqrt(5)): bound = 1/sqrt(fan_in)
        assert float(w.abs().max()) <= 1 / 24 ** 0.5 + 1e-6 and float(w.abs().max()) > 0.9 / 24 ** 0.5
        b = sd["encoder._blocks.2._se_reduce.bias"]
        assert 0 < float(b.abs().max()) <= 1 / 144 ** 0.5 + 1e-6
        m.load_state_dict(so)
        assert all(torch.equal(m.state_dict()[k], so[k]) for k in so)
        logical = sum(float(v.double().abs().sum()) for k, v in so.items() if v.dtype.is_floating_point and "running" not in k)
        assert abs(float(m.flat_parameters().double().abs().sum()) - logical) < 1e-6 * logical
        assert m.stages[0][0] == 0 and m.stages[-1][1] == m.flat_parameters().numel()
        assert m._n_mb == 32 and m._mb_drop[0] == 0 and abs(m._mb_drop[31] - 0.2 * 31 / 32) < 1e-7
        assert m._mb_drop[2] == 0                               # block 2 changes stride/width: no identity skip, no drop


def test_init_follows_smp_spec(U):
    torch.manual_seed(0)
    m = U.Unet("resnet18")
    sd = m.state_dict()
    w = sd["encoder.layer1.0.conv1.weight"]               # kaiming_normal fan_out: std = sqrt(2/(64*9))
    assert abs(float(w.std()) - (2.0 / (64 * 9)) ** 0.5) < 0.1 * (2.0 / (64 * 9)) ** 0.5
    wd = sd["decoder.blocks.0.conv1.0.weight"]            # kaiming_uniform fan_in: bound = sqrt(6/fan_in)
    assert float(wd.abs().max()) <= (6.0 / (768 * 9)) ** 0.5 + 1e-6
    assert float(sd["segmentation_head.0.bias"].abs().sum()) == 0.0
    assert torch.all(sd["encoder.bn1.weight"] == 1) and torch.all(sd["encoder.bn1.running_var"] == 1)
    assert int(sd["encoder.bn1.num_batches_tracked"]) == 0


def test_constructor_rejects_unsupported(U):
    for kw in (dict(encoder_name="mobilenet_v2"), dict(encoder_weights="imagenet"), dict(decoder_attention_type="scse"),
               dict(activation="sigmoid"), dict(aux_params={"classes": 2}), dict(decoder_channels=(256, 128, 64)),
               dict(encoder_depth=4, decoder_channels=(256, 128, 64, 32)), dict(decoder_channels=(256, 128, 64, 32, 10))):
        with pytest.raises(ValueError):
            U.Unet(**kw)
    with pytest.raises(ValueError, match="Unsupported model"):
        U.create_model("DeepLabV3Plus")
    with pytest.raises(ValueError):
        U.UnetPlusPlus(encoder_name="timm-regnety_016")


def test_config_factories(U):
    class NS:
        def __init__(self, **kw): self.__dict__.update(kw)
    cfg = NS(MODEL=NS(NAME="Unet", ENCODER_NAME="resnet18", ENCODER_WEIGHTS=None, IN_CHANNELS=3, CLASSES=1, ACTIVATION=None,
                      ENCODER_DEPTH=5, DECODER_CHANNELS=[256, 128, 64, 32, 16]),
             LOSS=NS(NAME="DiceLoss", MODE="binary", SMOOTH=1e-5, BCE_WEIGHT=0.5, DICE_WEIGHT=0.5))
    assert isinstance(U.create_model_from_config(cfg), U.Unet)
    cfg.MODEL.NAME = "UnetPlusPlus"                      # the reference's default MODEL.NAME
    mpp = U.create_model_from_config(cfg)
    assert isinstance(mpp, U.UnetPlusPlus)
    from oracle import unet_oracle as O
    ref = O.build("resnet18", arch="UnetPlusPlus")
    assert list(mpp.state_dict().keys()) == list(ref.state_dict().keys())
    assert all(mpp.state_dict()[k].shape == v.shape for k, v in ref.state_dict().items())
    mpp.load_state_dict(ref.state_dict())
    cfg.MODEL.NAME, cfg.MODEL.ENCODER_NAME = "Unet", "resnet50"       # unet_watermark_large.yaml's encoder
    m50 = U.create_model_from_config(cfg)
    r50 = O.build("resnet50")
    assert list(m50.state_dict().keys()) == list(r50.state_dict().keys())
    assert all(m50.state_dict()[k].shape == v.shape for k, v in r50.state_dict().items())
    assert isinstance(U.get_loss_function(cfg), U.DiceLoss)
    cfg.LOSS.NAME = "CombinedLoss"
    assert isinstance(U.get_loss_function(cfg), U.CombinedLoss)
    cfg.LOSS.NAME = "LovaszLoss"
    with pytest.raises(ValueError):
        U.get_loss_function(cfg)


def test_no_cpu_fallback(U):
    m = U.Unet("resnet18")
    with pytest.raises(RuntimeError, match="HIP device"):
        m(torch.zeros(1, 3, 64, 64))
    with pytest.raises(RuntimeError, match="HIP device"):
        U.DiceLoss()(torch.zeros(1, 1, 8, 8), torch.zeros(1, 1, 8, 8))
    with pytest.raises(RuntimeError, match="HIP device"):
        U.get_metrics()(torch.zeros(1, 8, 8), torch.zeros(1, 8, 8, dtype=torch.long))


def test_product_does_not_import_oracle():
    """The oracle is test infrastructure: nothing under the product package may import or execute it."""
    pkg = os.path.join(ROOT, "unet-watermark_amd")
    pat = re.compile(r"^\s*(from\s+oracle|import\s+oracle|from\s+\.+oracle)|oracle\.|unet_oracle", re.M)
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                src = open(os.path.join(dp, f)).read()
                assert not pat.search(src), f"{f} references the oracle"


def test_config_yaml_and_checkpoint_roundtrip(U, tmp_path):
    from unet_watermark_amd.config import get_cfg_defaults, update_config
    from unet_watermark_amd.checkpoint import save_checkpoint, load_checkpoint
    cfg = get_cfg_defaults()
    assert cfg.MODEL.ENCODER_NAME == "resnet34" and cfg.TRAIN.BATCH_SIZE == 16 and cfg.LOSS.SMOOTH == 1e-5
    y = tmp_path / "c.yaml"
    y.write_text("MODEL:\n  ENCODER_NAME: resnet18\nTRAIN:\n  LR: 0.005\nLOSS:\n  NAME: CombinedLoss\n  FOCAL_WEIGHT: 0.2\n")
    update_config(cfg, str(y))
    assert cfg.MODEL.ENCODER_NAME == "resnet18" and cfg.TRAIN.LR == 0.005 and cfg.LOSS.FOCAL_WEIGHT == 0.2
    assert cfg.MODEL.DECODER_CHANNELS == [256, 128, 64, 32, 16]          # untouched defaults survive the merge
    m = U.create_model_from_config(cfg)
    p = save_checkpoint(str(tmp_path / "ck" / "best.pth"), m, epoch=3, val_loss=0.5, val_metrics={"iou": 0.1}, cfg=cfg)
    m2 = U.create_model_from_config(cfg)
    ck = load_checkpoint(p, m2)
    assert ck["epoch"] == 3 and ck["config"]["MODEL"]["ENCODER_NAME"] == "resnet18"
    assert all(torch.equal(a, b) for a, b in zip(m.state_dict().values(), m2.state_dict().values()))
    torch.save(m.state_dict(), str(tmp_path / "bare.pth"))               # old format: bare state_dict
    assert load_checkpoint(str(tmp_path / "bare.pth"), m2)["epoch"] == 0


def test_cli_parses_reference_flags(U):
    from unet_watermark_amd import cli
    import pytest as _pt
    with _pt.raises(SystemExit):
        cli.main(["train", "--epochs", "1", "--batch-size", "2", "--lr", "0.001", "--no-early-stopping", "--synthetic", "8"])


def test_workspace_plan_sizes(U):
    """uwm_workspace_bytes is host-side planning: sizes must stay where DESIGN.md §3 says (a plan regression once
    allocated the decoder gradients at 8x the resolution) and scale ~linearly with the batch."""
    import ctypes as C
    from unet_watermark_amd import _lib as L
    lib = L.lib()
    gib = {}
    for arch in ("Unet", "UnetPlusPlus"):
        for enc in ("resnet34", "resnet50"):
            desc = L.uwm_unet_desc(L.ENC[enc], 3, 1, (C.c_int * 5)(256, 128, 64, 32, 16), 1e-5, 0.1, L.ARCH[arch])
            h = C.c_void_p()
            L.check(lib.uwm_create(C.byref(desc), C.byref(h)))
            b1, b16, b64 = (lib.uwm_workspace_bytes(h, n, 512, 512, 1) for n in (1, 16, 64))
            assert b16 < 16.5 * b1 and b64 < 4.1 * b16, (arch, enc, b1, b16, b64)
            assert lib.uwm_workspace_bytes(h, 16, 512, 512, 0) < b16
            gib[(arch, enc)] = b16 / 2 ** 30
            lib.uwm_destroy(h)
    desc = L.uwm_unet_desc(L.ENC["efficientnet-b4"], 3, 1, (C.c_int * 5)(256, 128, 64, 32, 16), 1e-5, 0.1, 0)
    h = C.c_void_p()
    L.check(lib.uwm_create(C.byref(desc), C.byref(h)))
    b1, b16 = (lib.uwm_workspace_bytes(h, n, 512, 512, 1) for n in (1, 16))
    assert b16 < 16.5 * b1 and b16 / 2 ** 30 < 40, (b1, b16)    # every MBConv stage materialised (round-1 path)
    lib.uwm_destroy(h)
    assert 6.5 < gib[("Unet", "resnet34")] < 8.5, gib           # BASELINE config 2: ~7.5 GB of the 288 GB
    assert gib[("UnetPlusPlus", "resnet34")] < 16 and gib[("UnetPlusPlus", "resnet50")] < 30, gib


def test_conv_flops_match_survey_and_oracle(U):
    """uwm_conv_flops (the roofline numerator bench.py uses, so the bench needs nothing from oracle/) = SURVEY 8(d)'s
    figures and the oracle's independent count, for every encoder / decoder the build serves."""
    from oracle import unet_oracle as O
    f, fb = U.Unet("resnet34").conv_flops(512, 512)
    assert round(f / 1e9, 3) == 62.512 and round(fb / 1e9, 3) == 186.303
    for enc, arch, hw in [("resnet18", "Unet", 256), ("resnet50", "Unet", 512), ("efficientnet-b4", "Unet", 1024),
                          ("resnet34", "UnetPlusPlus", 512)]:
        got = getattr(U, arch)(enc).conv_flops(hw, hw)
        ref = O.conv_flops(enc, hw, hw, arch=arch)
        assert all(abs(a - b) <= 1e-9 * b for a, b in zip(got, ref)), (enc, arch, got, ref)


def test_fused_optimizer_state_interchanges_with_torch_optim(U):
    """The fused optimizers save / load torch.optim's own state layout per parameter in logical (OIHW) shape
    (/root/reference/src/train.py:323-326,441-458 resumes `optimizer_state_dict` of optim.Adam): a torch Adam state loads
    into FusedAdam and back; an unknown layout is skipped with a warning instead of raising."""
    import warnings
    from unet_watermark_amd.train import FusedAdam, FusedSGD
    torch.manual_seed(3)
    m = U.Unet("resnet18")
    params = list(m.parameters())
    ref = torch.optim.Adam(params, lr=1e-3, weight_decay=1e-4)
    for p in params:
        p.grad = torch.randn_like(p) * 1e-2
    ref.step(); ref.step()
    sd = ref.state_dict()
    opt = FusedAdam(m, lr=5e-4)
    opt.load_state_dict(sd)
    assert opt._step == 2 and opt.param_groups[0]["lr"] == 1e-3 and opt.param_groups[0]["weight_decay"] == 1e-4
    views_m, views_v = opt._views(opt._bufs[0]), opt._views(opt._bufs[1])
    for i, p in enumerate(params):
        assert torch.equal(views_m[i], sd["state"][i]["exp_avg"]) and torch.equal(views_v[i], sd["state"][i]["exp_avg_sq"])
    # arena padding (rows padded to 32 floats) stays zero: the moments cover exactly the logical elements
    assert float(opt._bufs[0].abs().sum()) == pytest.approx(float(sum(s["exp_avg"].abs().sum() for s in sd["state"].values())), rel=1e-5)
    out = opt.state_dict()
    assert set(out) == {"state", "param_groups"} and len(out["state"]) == len(params)
    assert out["param_groups"][0]["params"] == list(range(len(params)))
    assert all(tuple(out["state"][i]["exp_avg"].shape) == tuple(p.shape) for i, p in enumerate(params))
    ref2 = torch.optim.Adam(params, lr=1.0)
    ref2.load_state_dict(out)                                   # torch accepts what we write
    assert torch.equal(ref2.state[params[5]]["exp_avg"], sd["state"][5]["exp_avg"])
    assert float(ref2.state[params[5]]["step"]) == 2.0
    # round-1 flat layout still loads; a foreign layout warns and leaves the state untouched
    opt2 = FusedAdam(m)
    opt2.load_state_dict({"step": 7, "exp_avg": opt._bufs[0].clone(), "exp_avg_sq": opt._bufs[1].clone(), "param_groups": [{"lr": 0.25}]})
    assert opt2._step == 7 and torch.equal(opt2._bufs[0], opt._bufs[0]) and opt2.param_groups[0]["lr"] == 0.25
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        opt2.load_state_dict({"foo": 1})
        FusedSGD(m).load_state_dict(sd)                         # Adam slots offered to SGD
    assert len(w) == 2 and opt2._step == 7
    sgd = torch.optim.SGD(params, lr=1e-2, momentum=0.9)
    sgd.step()
    fs = FusedSGD(m)
    fs.load_state_dict(sgd.state_dict())
    assert torch.equal(fs._views(fs._bufs[0])[3], sgd.state[params[3]]["momentum_buffer"])
    # frozen parameters are refused, not silently trained
    params[0].requires_grad_(False)
    with pytest.raises(NotImplementedError):
        opt._check_frozen()


def test_soft_dice_iou_and_early_stopping(U):
    """dice_coef / iou_score (/root/reference/src/utils/metrics.py:39-54) and the EarlyStopping rule
    (/root/reference/src/train.py:37-66) the CLI applies identically on every rank."""
    p = torch.tensor([[0.9, 0.1], [0.8, 0.0]]); t = torch.tensor([[1, 0], [1, 0]])
    i, sp, st = 1.7, 1.8, 2.0
    assert U.dice_coef(p, t) == pytest.approx((2 * i + 1e-5) / (sp + st + 1e-5), rel=1e-6)
    assert U.iou_score(p, t) == pytest.approx((i + 1e-5) / (sp + st - i + 1e-5), rel=1e-6)
    assert U.dice_coef(torch.zeros(4), torch.zeros(4)) == pytest.approx(1.0)
    from unet_watermark_amd.cli import EarlyStopping, _make_scheduler
    class M(torch.nn.Module):
        def __init__(self):
            super().__init__(); self.w = torch.nn.Parameter(torch.zeros(1))
    mm = M(); es = EarlyStopping(patience=2)
    seq = [1.0, 0.9, 0.95, 0.91]
    res = []
    for k, v in enumerate(seq):
        with torch.no_grad():
            mm.w.fill_(float(k))
        res.append(es(v, mm))
    assert res == [False, False, False, True] and float(mm.w) == 1.0      # stops after 2 bad epochs, restores the best weights
    from unet_watermark_amd.config import get_cfg_defaults
    cfg = get_cfg_defaults()
    assert cfg.MODEL.NAME == "UnetPlusPlus"                                # the reference's default (src/configs/config.py:15)
    o = torch.optim.SGD(mm.parameters(), lr=0.1)
    assert isinstance(_make_scheduler(cfg, o), torch.optim.lr_scheduler.ReduceLROnPlateau)
    cfg.OPTIMIZER.LR_SCHEDULER = "CosineAnnealingLR"; cfg.TRAIN.EPOCHS = 7
    sc = _make_scheduler(cfg, o)
    assert isinstance(sc, torch.optim.lr_scheduler.CosineAnnealingLR) and sc.T_max == 7
    cfg.OPTIMIZER.LR_SCHEDULER = "none"
    assert _make_scheduler(cfg, o) is None
