"""`Unet` — drop-in for `smp.Unet(...)` as the reference constructs it
(/root/reference/src/models/unet_model.py:64-71,93-120), executing on libuwm's HIP kernels.

* same constructor keywords as smp.Unet; unsupported values raise ValueError
  (mirroring unet_model.py:55-59's explicit rejection).
* parameters / BatchNorm buffers are real torch tensors that alias ONE flat device arena whose
  layout libuwm defines; `state_dict()` keys and logical OIHW shapes are smp-compatible
  (SURVEY.md Appendix A.2), so checkpoints interchange with the reference.
* forward/backward run only on a HIP device; there is no CPU or eager-PyTorch fallback.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Sequence

import torch
import torch.nn as nn

from . import _lib as L


class _Node(nn.Module):
    """Plain container used to reproduce smp's module tree (names only)."""


def _get_node(root: nn.Module, path: Sequence[str]) -> nn.Module:
    cur = root
    for name in path:
        nxt = cur._modules.get(name)
        if nxt is None:
            nxt = _Node()
            cur.add_module(name, nxt)
        cur = nxt
    return cur


class SegmentationModelError(RuntimeError):
    pass


class _UnetFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, module, x, *params):
        logits = module._forward_raw(x, training=True)
        ctx.module = module
        ctx.gen = module._fwd_gen
        return module._logits_view(logits)

    @staticmethod
    def backward(ctx, grad_out):
        m = ctx.module
        if ctx.gen != m._fwd_gen:
            raise RuntimeError("uwm: backward called after another forward overwrote the workspace "
                               "(one forward/backward pair at a time per model)")
        dl = m._as_padded_dlogits(grad_out)
        m._backward_raw(dl)
        if m._arena_grads_only:         # a fused flat optimizer consumes the arena itself (train.py): no per-parameter copies
            return (None, None) + (None,) * len(m._plist)
        grads = tuple(m._grad_views[i] if p.requires_grad else None for i, p in enumerate(m._plist))
        return (None, None) + grads


class Unet(nn.Module):
    """MI355X-native U-Net (ResNet-18/34/50 or EfficientNet-b4 encoder, smp UnetDecoder, 3x3 segmentation head)."""

    SUPPORTED_ENCODERS = tuple(L.ENC)
    _ARCH = "Unet"

    def __init__(self, encoder_name: str = "resnet34", encoder_depth: int = 5,
                 encoder_weights: Optional[str] = None, decoder_use_batchnorm: bool = True,
                 decoder_channels: Sequence[int] = (256, 128, 64, 32, 16),
                 decoder_attention_type: Optional[str] = None, in_channels: int = 3, classes: int = 1,
                 activation=None, aux_params: Optional[dict] = None):
        super().__init__()
        if encoder_name not in L.ENC:
            raise ValueError(f"Unsupported encoder: {encoder_name}. Supported encoders: {list(L.ENC)}")
        if encoder_depth != 5:
            raise ValueError(f"encoder_depth={encoder_depth} is not supported (only 5)")
        decoder_channels = tuple(int(c) for c in decoder_channels)
        if len(decoder_channels) != encoder_depth:
            raise ValueError(
                f"Model depth is {encoder_depth}, but you provide `decoder_channels` for {len(decoder_channels)} blocks.")
        if encoder_weights is not None:
            raise ValueError(
                f"encoder_weights={encoder_weights!r} needs a network download, which this build cannot do; "
                f"pass encoder_weights=None and load a local state_dict instead")
        if decoder_use_batchnorm is not True:
            raise ValueError("decoder_use_batchnorm must be True")
        if decoder_attention_type is not None:
            raise ValueError(f"decoder_attention_type={decoder_attention_type!r} is not supported")
        if activation is not None:
            raise ValueError(f"activation={activation!r} is not supported (the reference passes None)")
        if aux_params is not None:
            raise ValueError("aux_params (classification head) is not supported")

        self.encoder_name, self.in_channels, self.classes = encoder_name, int(in_channels), int(classes)
        self.decoder_channels = decoder_channels
        lib = L.lib()
        desc = L.uwm_unet_desc(L.ENC[encoder_name], self.in_channels, self.classes,
                               (C.c_int * 5)(*decoder_channels), 1e-5, 0.1, L.ARCH[self._ARCH])
        h = C.c_void_p()
        L.check(lib.uwm_create(C.byref(desc), C.byref(h)), ValueError)
        self._h = h
        self._cp = lib.uwm_logits_channels(h)
        self._n_param = lib.uwm_param_arena_floats(h)
        self._n_buf = lib.uwm_buffer_arena_floats(h)
        self._infos = []
        for i in range(lib.uwm_num_tensors(h)):
            ti = L.uwm_tensor_info()
            L.check(lib.uwm_tensor_info_get(h, i, C.byref(ti)))
            self._infos.append((ti.name.decode(), ti.kind, ti.arena, int(ti.offset),
                                tuple(ti.shape[: ti.ndim]), tuple(ti.stride[: ti.ndim])))
        self.stages = []
        for s in range(lib.uwm_num_stages(h)):
            b, e = C.c_longlong(), C.c_longlong()
            L.check(lib.uwm_stage_range(h, s, C.byref(b), C.byref(e)))
            self.stages.append((int(b.value), int(e.value)))

        self._param_arena = torch.zeros(self._n_param, dtype=torch.float32)
        self._buffer_arena = torch.zeros(self._n_buf, dtype=torch.float32)
        self._grad_arena = None
        self._plist, self._pinfo, self._binfo = [], [], []
        for name, kind, arena, off, shape, stride in self._infos:
            *path, leaf = name.split(".")
            node = _get_node(self, path)
            if arena == L.ARENA_PARAM:
                p = nn.Parameter(self._param_arena.as_strided(shape, stride, off))
                node.register_parameter(leaf, p)
                self._plist.append(p)
                self._pinfo.append((off, shape, stride))
            else:
                node.register_buffer(leaf, self._buffer_arena.as_strided(shape, stride, off))
                self._binfo.append((node, leaf, off, shape, stride))
                if kind == L.KIND_BN_VAR:
                    self._nbt = getattr(self, "_nbt", [])
                    self._nbt.append(node)
        # all num_batches_tracked counters alias one int64 arena: ONE increment kernel per training forward
        self._nbt_arena = torch.zeros(len(self._nbt), dtype=torch.long)
        for i, node in enumerate(self._nbt):
            node.register_buffer("num_batches_tracked", self._nbt_arena[i])
        self._grad_views = []
        self._ws = None
        self._ws_key = None
        self._fwd_gen = 0
        self._bound = None
        # EfficientNet encoders: stochastic depth of the MBConv blocks (efficientnet_pytorch drop_connect_rate 0.2)
        self._n_mb = lib.uwm_num_mbconv_blocks(h)
        self._mb_drop = [float(lib.uwm_mbconv_drop_rate(h, i)) for i in range(self._n_mb)]
        self.drop_connect = self._n_mb > 0
        self.precision = {v: k for k, v in L.PREC.items()}[lib.uwm_get_precision(h)]       # the process default (UWM_PRECISION) or "f32"
        self._fused_opt_ref = None        # weakref to the fused flat optimizer that consumes the gradient arena itself (train.py)
        self._keep_override = None        # tests: {0,1} keep masks [n_blocks, N] instead of a random draw
        self._rowscale = None
        self.reset_parameters()

    # ------------------------------------------------------------------ init
    @torch.no_grad()
    def reset_parameters(self):
        pi = 0
        for name, kind, arena, off, shape, stride in self._infos:
            if arena == L.ARENA_PARAM:
                p = self._plist[pi]; pi += 1
                if kind == L.KIND_CONV_W:
                    w = torch.empty(shape, dtype=torch.float32)
                    fan_in = shape[1] * shape[2] * shape[3]
                    if name.startswith("encoder._"):        # efficientnet_pytorch keeps torch's default Conv2d init
                        nn.init.kaiming_uniform_(w, a=5 ** 0.5)
                    elif name.startswith("encoder."):
                        nn.init.kaiming_normal_(w, mode="fan_out", nonlinearity="relu")
                    elif name.startswith("decoder."):
                        nn.init.kaiming_uniform_(w, mode="fan_in", nonlinearity="relu")
                    else:
                        nn.init.xavier_uniform_(w)
                    p.copy_(w.to(p.device))
                elif kind == L.KIND_BN_GAMMA:
                    p.fill_(1.0)
                elif kind == L.KIND_BIAS and name.startswith("encoder._"):   # default Conv2d bias init (SE layers)
                    bound = 1.0 / fan_in ** 0.5
                    p.copy_(torch.empty(shape, dtype=torch.float32).uniform_(-bound, bound).to(p.device))
                else:
                    p.zero_()
        for node, leaf, off, shape, stride in self._binfo:
            getattr(node, leaf).fill_(1.0 if leaf == "running_var" else 0.0)
        self._nbt_arena.zero_()

    # ------------------------------------------------------------------ device movement keeps the arena aliasing
    def _apply(self, fn, recurse=True):
        new_p = fn(self._param_arena)
        new_b = fn(self._buffer_arena)
        if new_p.dtype != torch.float32 or new_b.dtype != torch.float32:
            raise TypeError("uwm.Unet parameters are fp32 only (half()/bfloat16() are not supported)")
        self._param_arena, self._buffer_arena = new_p.contiguous(), new_b.contiguous()
        self._grad_arena = None
        self._grad_views = []
        for p, (off, shape, stride) in zip(self._plist, self._pinfo):
            p.data = self._param_arena.as_strided(shape, stride, off)
            p.grad = None
        for node, leaf, off, shape, stride in self._binfo:
            node._buffers[leaf] = self._buffer_arena.as_strided(shape, stride, off)
        self._nbt_arena = fn(self._nbt_arena)
        for i, node in enumerate(self._nbt):
            node._buffers["num_batches_tracked"] = self._nbt_arena[i]
        self._ws = None
        self._ws_key = None
        self._bound = None
        return self

    @property
    def device(self):
        return self._param_arena.device

    @property
    def _arena_grads_only(self) -> bool:
        """True while a LIVE fused flat optimizer (train.py) owns the gradient arena: backward() then leaves `p.grad` unset.
        The flag dies with that optimizer (or its close()), so a torch.optim optimizer, clip_grad_norm_ or a gradient
        check used on the same model afterwards sees populated `p.grad` again."""
        ref = self._fused_opt_ref
        return ref is not None and ref() is not None

    def flat_parameters(self) -> torch.Tensor:
        """The flat fp32 parameter arena (padding included; padding is always zero)."""
        return self._param_arena

    def flat_grads(self) -> torch.Tensor:
        self._ensure_bound()
        return self._grad_arena

    def num_parameters(self) -> int:
        return int(L.lib().uwm_param_count(self._h))

    def set_precision(self, mode: str = "f32", min_workgroups: Optional[int] = None, routing_batch: Optional[int] = None):
        """Arithmetic of the convolution products, per model (uwm_set_precision): "f32" (default: exact fp32 matrix
        instructions); "bf16x3" (opt-in: the backward data-gradient convolutions take each product as a_hi*b_hi + a_hi*b_lo +
        a_lo*b_hi over bf16 halves of the fp32 operands, fp32 accumulation — the forward, hence every logit, is unchanged);
        "bf16x3_all" (forward products too: 1.6e-3 logit error on resnet34, outside the 1e-3 bar on deep encoders);
        "f16x3" / "f16x3_all" (3x3 stride-1 convolutions — forward / forward, data and weight gradients — in direct form on
        gfx950's v_mfma_f32_16x16x32_f16 with every fp32 operand split into two fp16 halves: 22-bit operands, fp32 accumulation,
        power-of-two range scaling; fp32-class accuracy: the fp32 mode's parity bars hold, logit error 1.0e-4 either way).
        Parameters, activations, gradients and optimizer state stay fp32.  The reference's GPU path is reduced precision
        as well (fp16 autocast + GradScaler, /root/reference/src/train.py:75,89-98)."""
        if mode not in L.PREC:
            raise ValueError(f"unsupported precision {mode!r} (supported: {list(L.PREC)})")
        L.check(L.lib().uwm_set_precision(self._h, L.PREC[mode]), ValueError)
        if min_workgroups is not None:       # fp16x3 modes: smallest launch the fp16x3 kernels take (default: one workgroup per two CUs)
            L.check(L.lib().uwm_set_precision_fill(self._h, int(min_workgroups)), ValueError)
        if routing_batch is not None:        # choose kernels as if the batch were this many images (0: the real batch) — uwm_set_routing_batch
            L.check(L.lib().uwm_set_routing_batch(self._h, int(routing_batch)), ValueError)
        self.precision = mode
        return self

    def routing(self, enable: Optional[bool] = None, clear: bool = True):
        """Routing record of the library (uwm_routing_enable / uwm_routing_dump): with `enable` switch the record on / off; otherwise
        return the list of (pass, layer, kernel) of every convolution-class launch since the last call — which HIP kernel each
        layer's forward / dgrad / wgrad ran on (tests and bench.py assert it for the benched precision mode)."""
        if enable is not None:
            L.check(L.lib().uwm_routing_enable(self._h, int(bool(enable))))
            return self
        need = int(L.lib().uwm_routing_dump(self._h, None, 0, 0))
        buf = C.create_string_buffer(max(1, need))
        L.lib().uwm_routing_dump(self._h, buf, need, int(clear))
        out = []
        for line in buf.value.decode().splitlines():
            if line:
                pas, layer, kern = line.split(" ", 2)
                out.append((pas, layer, kern.strip("()")))       # (a templated kernel is written "(name<args>)" at its launch site)
        return out

    def conv_flops(self, h: int, w: int):
        """Algorithmic conv FLOPs per image at h x w: (forward, forward+backward) — SURVEY.md 8(d)."""
        f, fb = C.c_double(), C.c_double()
        L.check(L.lib().uwm_conv_flops(self._h, int(h), int(w), C.byref(f), C.byref(fb)))
        return f.value, fb.value

    # ------------------------------------------------------------------ raw entry points
    def _require_gpu(self, x: Optional[torch.Tensor] = None):
        if self._param_arena.device.type != "cuda":
            raise RuntimeError("uwm.Unet runs only on a HIP device (no CPU fallback): call .to('cuda') first")
        if x is not None and x.device != self._param_arena.device:
            raise RuntimeError(f"input is on {x.device} but the model is on {self._param_arena.device}")

    def _ensure_bound(self):
        self._require_gpu()
        if self._grad_arena is None:
            self._grad_arena = torch.zeros_like(self._param_arena)
            self._grad_views = [self._grad_arena.as_strided(shape, stride, off) for off, shape, stride in self._pinfo]
            self._bound = None
        key = (self._param_arena.data_ptr(), self._grad_arena.data_ptr(), self._buffer_arena.data_ptr())
        if self._bound != key:
            L.check(L.lib().uwm_bind(self._h, C.c_void_p(key[0]), C.c_void_p(key[1]), C.c_void_p(key[2])))
            self._bound = key

    def _workspace(self, n, h, w, training):
        key = (n, h, w, bool(training))
        need = L.lib().uwm_workspace_bytes(self._h, n, h, w, int(training))
        if need == 0:
            raise SegmentationModelError(L.lib().uwm_last_error().decode())
        if self._ws is None or self._ws.numel() < need or self._ws.device != self.device:
            self._ws = torch.empty(need, dtype=torch.uint8, device=self.device)
        self._ws_key = key
        return self._ws

    def check_input_shape(self, x):
        h, w = x.shape[-2:]
        if h % 32 != 0 or w % 32 != 0:
            nh = (h // 32 + 1) * 32 if h % 32 else h
            nw = (w // 32 + 1) * 32 if w % 32 else w
            raise RuntimeError(
                f"Wrong input shape height={h}, width={w}. Expected image height and width divisible by 32. "
                f"Consider pad your images to shape ({nh}, {nw}).")

    def _forward_raw(self, x: torch.Tensor, training: bool) -> torch.Tensor:
        """x [N,Cin,H,W] fp32 on the HIP device -> padded logits [N,H,W,CP]."""
        self._require_gpu(x)
        if x.dim() != 4 or x.shape[1] != self.in_channels:
            raise RuntimeError(f"expected input of shape (N,{self.in_channels},H,W), got {tuple(x.shape)}")
        self.check_input_shape(x)
        if x.dtype != torch.float32:
            raise TypeError(f"uwm.Unet takes float32 images, got {x.dtype}")
        x = x.contiguous()
        n, _, h, w = x.shape
        self._ensure_bound()
        ws = self._workspace(n, h, w, training)
        logits = torch.empty((n, h, w, self._cp), dtype=torch.float32, device=x.device)
        self._fwd_gen += 1
        if self._n_mb:
            self._set_drop_connect(n, training, x.device)
        L.check(L.lib().uwm_forward(self._h, C.c_void_p(x.data_ptr()), C.c_void_p(logits.data_ptr()),
                                    C.c_void_p(ws.data_ptr()), ws.numel(), n, h, w, int(training),
                                    C.c_void_p(L.stream_ptr(x.device))), SegmentationModelError)
        if training:
            self._nbt_arena += 1
        return logits

    def _set_drop_connect(self, n: int, training: bool, device):
        """Draw this step's per-block, per-sample keep masks (efficientnet_pytorch utils.drop_connect: floor(keep_prob +
        U[0,1)) / keep_prob) on the device and hand them to the library; the tensor lives until the next forward."""
        if not (training and self.drop_connect):
            self._rowscale = None
            L.check(L.lib().uwm_set_drop_connect(self._h, None))
            return
        kp = getattr(self, "_keep_prob", None)
        if kp is None or kp.device != device:           # (cached: a host-to-device copy has no place in a captured step)
            kp = self._keep_prob = 1.0 - torch.tensor(self._mb_drop, dtype=torch.float32, device=device).view(-1, 1)
        keep_prob = kp
        if self._keep_override is not None:
            keep = self._keep_override.to(device=device, dtype=torch.float32).reshape(self._n_mb, n)
        else:
            keep = torch.floor(keep_prob + torch.rand((self._n_mb, n), dtype=torch.float32, device=device))
        self._rowscale = (keep / keep_prob).contiguous()
        L.check(L.lib().uwm_set_drop_connect(self._h, C.c_void_p(self._rowscale.data_ptr())))

    def _backward_raw(self, dlogits: torch.Tensor, stage_begin: int = 0, stage_end: Optional[int] = None):
        """dlogits [N,H,W,CP] (padding channels zero) -> gradient arena (overwritten)."""
        if stage_end is None:
            stage_end = len(self.stages)
        L.check(L.lib().uwm_backward(self._h, C.c_void_p(dlogits.data_ptr()), C.c_void_p(self._ws.data_ptr()),
                                     stage_begin, stage_end, C.c_void_p(L.stream_ptr(dlogits.device))))

    def debug_buffer(self, key: str) -> torch.Tensor:
        """Flat fp32 view of a planned workspace buffer (see uwm_debug_lookup) — for parity tests."""
        off, cnt = C.c_longlong(), C.c_longlong()
        L.check(L.lib().uwm_debug_lookup(self._h, key.encode(), C.byref(off), C.byref(cnt)))
        return self._ws.view(torch.float32)[off.value: off.value + cnt.value]

    def _logits_view(self, logits_nhwc: torch.Tensor) -> torch.Tensor:
        return logits_nhwc[..., : self.classes].permute(0, 3, 1, 2)

    def _as_padded_dlogits(self, g: torch.Tensor) -> torch.Tensor:
        n, c, h, w = g.shape
        cp = self._cp
        gp = g.permute(0, 2, 3, 1)
        if (getattr(g, "_uwm_padded", False) or g.data_ptr() in _PADDED_PTRS) and gp.stride()[:3] == (h * w * cp, w * cp, cp) \
                and (c == 1 or gp.stride(3) == 1):
            return gp
        dl = torch.zeros((n, h, w, cp), dtype=torch.float32, device=g.device)
        dl[..., :c] = gp
        return dl

    # ------------------------------------------------------------------ nn.Module protocol
    def forward(self, x: torch.Tensor) -> torch.Tensor:
        needs_grad = torch.is_grad_enabled() and self.training and any(p.requires_grad for p in self._plist)
        if needs_grad:
            return _UnetFunction.apply(self, x, *self._plist)
        return self._logits_view(self._forward_raw(x, training=self.training))

    @torch.no_grad()
    def predict(self, x):
        if self.training:
            self.eval()
        return self.forward(x)

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                L.lib().uwm_destroy(self._h)
                self._h = None
        except Exception:
            pass


# data_ptrs of dlogits buffers whose padding channels are known to be zero (written by uwm_loss)
_PADDED_PTRS: set = set()


class UnetPlusPlus(Unet):
    """MI355X-native UNet++ (smp.UnetPlusPlus: ResNet-18/34 encoder, dense x_{depth}_{layer} decoder grid) — the
    reference's default architecture (MODEL.NAME, /root/reference/src/configs/config.py:15).  Same constructor
    surface, arenas, kernels and training path as `Unet`; state_dict keys `decoder.blocks.x_0_0.conv1.0.weight` ..."""
    _ARCH = "UnetPlusPlus"


# ---------------------------------------------------------------------------- factory (reference L3 glue)
SUPPORTED_MODELS = {"Unet": Unet, "UnetPlusPlus": UnetPlusPlus}


def create_model(model_name: str, encoder_name: str = "resnet34", encoder_weights: Optional[str] = None,
                 in_channels: int = 3, classes: int = 1, activation=None, **kwargs) -> nn.Module:
    """Counterpart of SMPModelFactory.create_model (/root/reference/src/models/unet_model.py:30-73)."""
    if model_name not in SUPPORTED_MODELS:
        raise ValueError(f"Unsupported model: {model_name}. Supported models: {list(SUPPORTED_MODELS.keys())}")
    return SUPPORTED_MODELS[model_name](encoder_name=encoder_name, encoder_weights=encoder_weights,
                                        in_channels=in_channels, classes=classes, activation=activation, **kwargs)


def create_model_from_config(cfg) -> nn.Module:
    """Counterpart of create_model_from_config (/root/reference/src/models/unet_model.py:93-120)."""
    m = cfg.MODEL
    params = dict(model_name=m.NAME, encoder_name=m.ENCODER_NAME, encoder_weights=m.ENCODER_WEIGHTS,
                  in_channels=m.IN_CHANNELS, classes=m.CLASSES, activation=m.ACTIVATION)
    if hasattr(m, "ENCODER_DEPTH"):
        params["encoder_depth"] = m.ENCODER_DEPTH
    if hasattr(m, "DECODER_CHANNELS"):
        params["decoder_channels"] = m.DECODER_CHANNELS
    return create_model(**params)
