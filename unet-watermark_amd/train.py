"""Training step of the hot path: forward -> Dice/BCE loss -> staged backward (gradient buckets
all-reduced over RCCL while the encoder backward is still running) -> fused Adam.

Counterpart of the five hot lines of /root/reference/src/train.py (:86,:91-98 / :100-105) plus the
data-parallel exchange the reference lacks: one process per GPU, replicas of
all parameters, per-rank BatchNorm statistics and per-rank loss — i.e. what wrapping the
reference in DistributedDataParallel would compute — with `torch.distributed` (backend "nccl" is
RCCL over xGMI) carrying one all-reduce per backward stage on a side HIP stream.
"""
from __future__ import annotations

import ctypes as C
import weakref
from typing import Optional, Sequence, Tuple

import torch
import torch.distributed as dist

from . import _lib as L


class GradReducer:
    """Bucketed gradient all-reduce over a flat gradient arena.

    `buckets` are [begin,end) element ranges (model.stages: head+decoder, layer4, layer3, layer2,
    layer1+stem — the order backward completes them).  reduce(k) enqueues the SUM all-reduce of
    bucket k on a side stream gated by an event recorded on the compute stream; finish() makes the
    compute stream wait for all of them.  Averaging (1/world) is folded into the optimizer's
    grad_scale.  Works on CPU tensors with the gloo backend (tests) — then without streams.
    """

    def __init__(self, flat_grads: torch.Tensor, buckets: Sequence[Tuple[int, int]], group=None, force: bool = False):
        self.flat = flat_grads
        self.buckets = [(int(b), int(e)) for b, e in buckets if e > b]
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1
        # force: run the collectives even in a 1-rank group (exercises the RCCL + side-stream path on one GPU)
        self.active = self.world > 1 or (force and dist.is_available() and dist.is_initialized())
        self.on_gpu = flat_grads.device.type == "cuda"
        self.comm_stream = torch.cuda.Stream(device=flat_grads.device) if (self.on_gpu and self.active) else None
        self._works = []

    def reduce(self, k: int):
        if not self.active:
            return
        b, e = self.buckets[k]
        view = self.flat[b:e]
        if self.on_gpu:
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(self.flat.device))
            self.comm_stream.wait_event(ev)
            with torch.cuda.stream(self.comm_stream):
                self._works.append(dist.all_reduce(view, op=dist.ReduceOp.SUM, group=self.group, async_op=True))
        else:
            self._works.append(dist.all_reduce(view, op=dist.ReduceOp.SUM, group=self.group, async_op=True))

    def finish(self):
        for w in self._works:
            w.wait()                       # GPU: current stream waits on the collective's stream
        self._works.clear()
        if self.comm_stream is not None:
            torch.cuda.current_stream(self.flat.device).wait_stream(self.comm_stream)


def broadcast_model(model, src: int = 0, group=None):
    """Rank `src`'s parameters and BatchNorm buffers to every rank (DDP's initial sync)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return
    dist.broadcast(model.flat_parameters(), src=src, group=group)
    dist.broadcast(model._buffer_arena, src=src, group=group)


class _FusedFlatOptimizer(torch.optim.Optimizer):
    """Base of the fused optimizers: ONE kernel launch over the model's flat parameter arena, reading the flat
    GRADIENT arena the last backward wrote.

    What that implies (and differs from a per-parameter torch optimizer):
    * the gradient arena is overwritten, not accumulated, by every backward, and `p.grad` is never read: gradient
      accumulation over several backward() calls is not available on this path (use torch.optim.* on
      model.parameters() with the autograd path for that);
    * every parameter is updated: a frozen parameter (requires_grad=False) raises instead of being silently trained;
    * by default the autograd path stops materialising `p.grad` (a copy of the whole arena per step that nothing on this
      path reads); pass materialize_grads=True to keep `p.grad` populated.
    Checkpoints: state_dict() / load_state_dict() use torch.optim's own layout ({'state': {i: {...}}, 'param_groups'}),
    per parameter and in logical (OIHW) shape, so a reference periodic checkpoint (/root/reference/src/train.py:441-458)
    resumes here and ours resumes there; a layout it does not recognise is skipped with a warning (weights only)."""

    _SLOTS = ()          # names of the per-parameter state tensors, in arena order

    def __init__(self, model, defaults, max_grad_norm=None, materialize_grads=False):
        self.model = model
        super().__init__(list(model.parameters()), defaults)
        self.max_grad_norm = max_grad_norm           # TRAIN.GRADIENT_CLIP (config.py:54; unused by the reference's train.py)
        self._clip_scratch = None
        self._bufs = None
        self._step = 0
        self._checked_flags = None
        # the autograd path skips the per-parameter gradient copies only while THIS optimizer is alive (weak reference):
        # dropping it, or close(), gives `p.grad` back to torch.optim / clip_grad_norm_ / gradient checks on the same model
        model._fused_opt_ref = None if materialize_grads else weakref.ref(self)

    def close(self):
        """Give the model's autograd path its per-parameter `p.grad` back (undo of materialize_grads=False)."""
        ref = getattr(self.model, "_fused_opt_ref", None)
        if ref is not None and ref() is self:
            self.model._fused_opt_ref = None

    def _state_init(self):
        p = self.model.flat_parameters()
        if self._bufs is None or self._bufs[0].device != p.device:
            old = self._bufs
            self._bufs = [torch.zeros_like(p) for _ in self._SLOTS]
            if old is not None:
                for n, o in zip(self._bufs, old):
                    n.copy_(o)

    def _check_frozen(self):
        flags = tuple(p.requires_grad for p in self.model.parameters())
        if flags != self._checked_flags:
            if not all(flags):
                names = [n for n, p in self.model.named_parameters() if not p.requires_grad]
                raise NotImplementedError(
                    f"{type(self).__name__} updates the whole flat arena and cannot skip frozen parameters "
                    f"({len(names)} have requires_grad=False, e.g. {names[0]}); use torch.optim on model.parameters()")
            self._checked_flags = flags

    def _clip_ptr(self, p):
        if not self.max_grad_norm:
            return None
        if self._clip_scratch is None or self._clip_scratch.device != p.device:
            self._clip_scratch = torch.zeros(2, dtype=torch.float64, device=p.device)
        return C.c_void_p(self._clip_scratch.data_ptr())

    def zero_grad(self, set_to_none: bool = True):
        # gradients live in the flat arena and are overwritten (not accumulated) by every backward
        for p in self.model.parameters():
            p.grad = None

    # ---- torch.optim-compatible state
    def _views(self, flat):
        return [flat.as_strided(shape, stride, off) for off, shape, stride in self.model._pinfo]

    def state_dict(self):
        self._state_init()
        n = len(self.model._pinfo)
        per = [self._views(b) for b in self._bufs]
        state = {}
        if self._step > 0:
            for i in range(n):
                ent = {"step": torch.tensor(float(self._step))}
                for name, views in zip(self._SLOTS, per):
                    ent[name] = views[i].detach().clone().contiguous()
                state[i] = ent
        groups = [dict({k: v for k, v in g.items() if k != "params"}, params=list(range(n))) for g in self.param_groups]
        return {"state": state, "param_groups": groups}

    def load_state_dict(self, sd):
        import warnings
        self._state_init()
        if "state" in sd and "param_groups" in sd:                      # torch.optim layout (ours and the reference's)
            st = sd["state"]
            n = len(self.model._pinfo)
            per = [self._views(b) for b in self._bufs]
            if len(st) not in (0, n) or any(nm not in ent for ent in st.values() for nm in self._SLOTS):
                warnings.warn(f"{type(self).__name__}: optimizer state has {len(st)} entries / other slots than "
                              f"{self._SLOTS} for {n} parameters; optimizer state NOT restored (weights only)")
                return
            for b in self._bufs:
                b.zero_()
            step = 0
            for i, ent in st.items():
                i = int(i)
                for name, views in zip(self._SLOTS, per):
                    t = ent[name]
                    if tuple(t.shape) != tuple(views[i].shape):
                        warnings.warn(f"{type(self).__name__}: state {name}[{i}] has shape {tuple(t.shape)}, expected "
                                      f"{tuple(views[i].shape)}; optimizer state NOT restored (weights only)")
                        for b in self._bufs:
                            b.zero_()
                        self._step = 0
                        return
                    views[i].copy_(t.to(views[i].device))
                step = max(step, int(float(ent.get("step", 0))))
            # torch.optim.SGD keeps no 'step': a loaded momentum buffer must not be taken for an uninitialised one (the
            # kernel's first step overwrites the buffer with the raw gradient), so restored state counts as >= 1 step
            self._step = max(step, 1) if len(st) else step
        elif "step" in sd and all(nm in sd for nm in self._SLOTS):      # round-1 flat-arena layout
            self._step = int(sd["step"])
            for b, nm in zip(self._bufs, self._SLOTS):
                if sd[nm] is not None and sd[nm].numel() == b.numel():
                    b.copy_(sd[nm].to(b.device))
        else:
            warnings.warn(f"{type(self).__name__}: unrecognised optimizer state layout (keys {sorted(sd)[:6]}); "
                          f"optimizer state NOT restored (weights only)")
            return
        for g, s_ in zip(self.param_groups, sd.get("param_groups", [])):
            g.update({k: v for k, v in s_.items() if k != "params"})


class FusedAdam(_FusedFlatOptimizer):
    """torch.optim.Adam(lr, betas, eps, weight_decay) semantics (coupled L2) as ONE kernel launch
    over the model's flat parameter arena (/root/reference/src/train.py:266-270 builds optim.Adam)."""

    _SLOTS = ("exp_avg", "exp_avg_sq")

    def __init__(self, model, lr=1e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, max_grad_norm=None,
                 materialize_grads=False):
        super().__init__(model, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay), max_grad_norm,
                         materialize_grads)

    @property
    def _m(self):
        return self._bufs[0] if self._bufs else None

    @property
    def _v(self):
        return self._bufs[1] if self._bufs else None

    # ---- hipGraph-capturable form: every hyper-parameter (and the step count) lives in device memory
    def _hyper_values(self, grad_scale: float):
        g = self.param_groups[0]
        return (float(g["lr"]), float(g["betas"][0]), float(g["betas"][1]), float(g["eps"]), float(g["weight_decay"]),
                float(grad_scale), float(self.max_grad_norm or 0.0))

    def sync_hyper(self, grad_scale: float = 1.0):
        """(Re)write the device-side hyper-parameter block of step_graph() from param_groups / the host step count.  Call
        OUTSIDE a capture; needed only when something changed (an LR scheduler step, a restored checkpoint)."""
        p = self.model.flat_parameters()
        vals = self._hyper_values(grad_scale)
        if getattr(self, "_hyper", None) is None or self._hyper.device != p.device:
            self._hyper = torch.zeros(10, dtype=torch.float32, device=p.device)
            self._hyper_host = None
        want = vals + (float(self._step),)
        if self._hyper_host != want:
            self._hyper.copy_(torch.tensor(want + (0.0, 0.0), dtype=torch.float32))
            self._hyper_host = want

    @torch.no_grad()
    def step_graph(self, grad_scale: float = 1.0):
        """One Adam update through uwm_adam_graph: no host-side value enters the launch, so it can sit in a captured
        hipGraph and be replayed for every step.  The caller mirrors the step count on the host (note_graph_step)."""
        self._state_init()
        self._check_frozen()
        p = self.model.flat_parameters()
        gr = self.model.flat_grads()
        if getattr(self, "_hyper", None) is None:
            raise RuntimeError("FusedAdam.step_graph: call sync_hyper() first (outside the capture)")
        with L.on_device(p):
            L.check(L.lib().uwm_adam_graph(C.c_void_p(p.data_ptr()), C.c_void_p(gr.data_ptr()), C.c_void_p(self._bufs[0].data_ptr()),
                                           C.c_void_p(self._bufs[1].data_ptr()), p.numel(), C.c_void_p(self._hyper.data_ptr()),
                                           self._clip_ptr(p), C.c_void_p(L.stream_ptr(p.device))))

    def note_graph_step(self):
        """Host mirror of the device step counter after one replay / eager step_graph()."""
        self._step += 1
        if getattr(self, "_hyper_host", None) is not None:
            self._hyper_host = self._hyper_host[:-1] + (float(self._step),)

    @torch.no_grad()
    def step(self, closure=None, grad_scale: float = 1.0):
        loss = closure() if closure is not None else None
        g = self.param_groups[0]
        self._state_init()
        self._check_frozen()
        self._step += 1
        p = self.model.flat_parameters()
        gr = self.model.flat_grads()
        args = (C.c_void_p(p.data_ptr()), C.c_void_p(gr.data_ptr()), C.c_void_p(self._bufs[0].data_ptr()),
                C.c_void_p(self._bufs[1].data_ptr()), p.numel(), float(g["lr"]), float(g["betas"][0]), float(g["betas"][1]),
                float(g["eps"]), float(g["weight_decay"]), self._step, float(grad_scale))
        with L.on_device(p):
            if self.max_grad_norm:
                L.check(L.lib().uwm_adam_clip(*args, float(self.max_grad_norm), self._clip_ptr(p),
                                              C.c_void_p(L.stream_ptr(p.device))))
            else:
                L.check(L.lib().uwm_adam(*args, C.c_void_p(L.stream_ptr(p.device))))
        return loss


class FusedSGD(_FusedFlatOptimizer):
    """torch.optim.SGD(lr, momentum=0.9, weight_decay) — the reference's OPTIMIZER.NAME == "SGD" branch
    (/root/reference/src/train.py:272-278) — as one launch over the flat arena (uwm_sgd)."""

    _SLOTS = ("momentum_buffer",)

    def __init__(self, model, lr=1e-4, momentum=0.9, weight_decay=0.0, max_grad_norm=None, materialize_grads=False):
        super().__init__(model, dict(lr=lr, momentum=momentum, weight_decay=weight_decay, dampening=0.0, nesterov=False),
                         max_grad_norm, materialize_grads)

    @torch.no_grad()
    def step(self, closure=None, grad_scale: float = 1.0):
        loss = closure() if closure is not None else None
        g = self.param_groups[0]
        self._state_init()
        self._check_frozen()
        self._step += 1
        p = self.model.flat_parameters()
        gr = self.model.flat_grads()
        with L.on_device(p):
            L.check(L.lib().uwm_sgd(C.c_void_p(p.data_ptr()), C.c_void_p(gr.data_ptr()), C.c_void_p(self._bufs[0].data_ptr()),
                                    p.numel(), float(g["lr"]), float(g["momentum"]), float(g["weight_decay"]), self._step,
                                    float(grad_scale), float(self.max_grad_norm or 0.0), self._clip_ptr(p),
                                    C.c_void_p(L.stream_ptr(p.device))))
        return loss


class Trainer:
    """The fused train step used by bench.py and the CLI: no autograd graph, no per-parameter
    Python work; same arithmetic as `model(x); criterion(...); loss.backward(); optimizer.step()`."""

    def __init__(self, model, w_dice: float = 1.0, w_bce: float = 0.0, smooth: float = 1e-5, eps: float = 1e-7,
                 lr: float = 1e-4, betas=(0.9, 0.999), adam_eps: float = 1e-8, weight_decay: float = 0.0,
                 group=None, overlap_comm: bool = True, force_ddp: bool = False, max_grad_norm=None,
                 optimizer: str = "Adam", momentum: float = 0.9, global_dice: bool = False, use_graph: bool = False):
        self.model = model
        # use_graph: the whole step (forward, loss, staged backward with its weight-gradient side stream, Adam) is captured
        # ONCE per batch shape into a hipGraph and replayed ("hipGraph the step"): the ~150-1200 launches of
        # a step cost the host one graph launch.  Single-process Adam only; a data-parallel trainer keeps the eager path
        # (its all-reduces are issued by torch.distributed between the backward stages).
        self.use_graph = bool(use_graph)
        self._graphs = {}
        # global_dice (SURVEY.md 8(e) caveat): Dice is a ratio of batch sums, so DDP's mean of per-rank Dice losses is not the
        # Dice of the global batch.  With the flag the four loss sums (32 bytes) are all-reduced between the loss's two
        # halves and every rank back-propagates the GLOBAL loss: an N-rank run then optimises exactly what one process
        # with the concatenated batch would (BatchNorm statistics stay per rank, as under torch DDP).
        self.global_dice = bool(global_dice)
        self.w_dice, self.w_bce, self.smooth, self.eps = float(w_dice), float(w_bce), float(smooth), float(eps)
        if optimizer == "Adam":          # cfg.OPTIMIZER.NAME (/root/reference/src/train.py:266-279)
            self.opt = FusedAdam(model, lr=lr, betas=betas, eps=adam_eps, weight_decay=weight_decay, max_grad_norm=max_grad_norm)
        elif optimizer == "SGD":
            self.opt = FusedSGD(model, lr=lr, momentum=momentum, weight_decay=weight_decay, max_grad_norm=max_grad_norm)
        else:
            raise ValueError(f"unsupported optimizer: {optimizer!r} (Adam | SGD)")
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1
        self.overlap = overlap_comm
        self.ddp = self.world > 1 or (force_ddp and dist.is_available() and dist.is_initialized())
        self._force = force_ddp
        self._reducer = None
        self._dl = None
        self._scratch = None
        self._loss = None
        self._shape_bufs = {}
        if self.world > 1:
            broadcast_model(model, 0, group)

    def _buffers(self, n, h, w, dev):
        # one (dlogits, loss scratch, loss) set PER batch shape, kept for the trainer's lifetime: a captured step (use_graph) has their
        # addresses baked in, so a second shape — a partial last batch — must not hand the first shape's blocks back to the allocator
        cp = self.model._cp
        key = (n, h, w, cp, dev)
        ent = self._shape_bufs.get(key)
        if ent is None:
            ent = (torch.empty((n, h, w, cp), dtype=torch.float32, device=dev), torch.empty(8, dtype=torch.float64, device=dev),
                   torch.empty(3, dtype=torch.float32, device=dev))
            self._shape_bufs[key] = ent
        self._dl, self._scratch, self._loss = ent

    def step(self, images: torch.Tensor, masks: torch.Tensor) -> torch.Tensor:
        """images (N,C,H,W) fp32, masks (N,H,W)|(N,1,H,W) int64|uint8|float32 on the HIP device.
        Returns a device tensor {total, dice, bce} (no host sync)."""
        if self.use_graph and not self.ddp and isinstance(self.opt, FusedAdam):
            return self._step_graph(images, masks)
        return self._step_eager(images, masks)

    def _step_graph(self, images: torch.Tensor, masks: torch.Tensor) -> torch.Tensor:
        key = (tuple(images.shape), tuple(masks.shape), masks.dtype, images.device)
        ent = self._graphs.get(key)
        if ent is None:
            # the first step of a shape runs eagerly (plans the workspace, sets kernel attributes, leaves no side-stream work
            # un-joined), then the step is captured for the ones that follow — a capture executes nothing
            out = self._step_eager(images, masks).clone()
            gx, gt = images.clone(), masks.clone()
            torch.cuda.synchronize(images.device)
            self.opt.sync_hyper(1.0 / self.world)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                loss = self._step_eager(gx, gt, graph_opt=True)
            # every buffer whose address the capture baked in is OWNED by the graph entry: the static input / target copies, the loss
            # buffers of this shape (self._shape_bufs keeps them too) and the model's workspace block as it was at capture time — a
            # later forward that needs more bytes (another shape, an eval batch) makes the model allocate a NEW block, and this
            # reference keeps the captured one alive instead of letting the caching allocator hand it to another tensor
            self._graphs[key] = (g, gx, gt, loss, (self._dl, self._scratch, self.model._ws))
            return out
        g, gx, gt, loss, _owned = ent
        self.opt.sync_hyper(1.0 / self.world)           # (a host compare; writes only after an LR change / restore)
        gx.copy_(images); gt.copy_(masks)
        g.replay()
        self.opt.note_graph_step()
        return loss

    def _step_eager(self, images: torch.Tensor, masks: torch.Tensor, graph_opt: bool = False) -> torch.Tensor:
        m = self.model
        if not m.training:
            m.train()
        n, _, h, w = images.shape
        dev = images.device
        logits = m._forward_raw(images, training=True)
        self._buffers(n, h, w, dev)
        t = masks.contiguous()
        if t.numel() != n * h * w:
            raise ValueError(f"masks have {t.numel()} elements, expected {n * h * w}")
        st = C.c_void_p(L.stream_ptr(dev))
        if self.global_dice and self.ddp:
            lp, tp, tdt = C.c_void_p(logits.data_ptr()), C.c_void_p(t.data_ptr()), L.target_dtype_code(t)
            L.check(L.lib().uwm_loss_sums(lp, m._cp, tp, tdt, n * h * w, C.c_void_p(self._scratch.data_ptr()), st))
            dist.all_reduce(self._scratch[:4], group=self.group)          # SUM of {sum p*t, sum p, sum t, sum bce}
            # the ranks' parameter gradients of the global loss ADD UP; the exchange below averages, hence grad_scale = world
            L.check(L.lib().uwm_loss_apply(lp, m._cp, tp, tdt, n * h * w, n * h * w * self.world, self.w_dice, self.w_bce,
                                           self.smooth, self.eps, C.c_void_p(self._scratch.data_ptr()),
                                           C.c_void_p(self._loss.data_ptr()), C.c_void_p(self._dl.data_ptr()), m._cp,
                                           float(self.world), st))
        else:
            L.check(L.lib().uwm_loss(C.c_void_p(logits.data_ptr()), m._cp, C.c_void_p(t.data_ptr()), L.target_dtype_code(t),
                                     n * h * w, self.w_dice, self.w_bce, self.smooth, self.eps,
                                     C.c_void_p(self._scratch.data_ptr()), C.c_void_p(self._loss.data_ptr()),
                                     C.c_void_p(self._dl.data_ptr()), m._cp, 1.0, st))
        nst = len(m.stages)
        if self.ddp:
            if self._reducer is None or self._reducer.flat.data_ptr() != m.flat_grads().data_ptr():
                self._reducer = GradReducer(m.flat_grads(), m.stages, self.group, force=self._force)
                if self.overlap and self._reducer.comm_stream is not None:
                    # each stage's weight gradients (library side stream) gate that stage's all-reduce on the
                    # communication stream instead of stalling the compute stream at every stage boundary
                    L.check(L.lib().uwm_set_join_stream(m._h, C.c_void_p(self._reducer.comm_stream.cuda_stream)))
            if self.overlap:
                for k in range(nst):
                    m._backward_raw(self._dl, k, k + 1)
                    self._reducer.reduce(k)
            else:
                m._backward_raw(self._dl, 0, nst)
                for k in range(nst):
                    self._reducer.reduce(k)
            self._reducer.finish()
        else:
            m._backward_raw(self._dl, 0, nst)
        if graph_opt:
            self.opt.step_graph(grad_scale=1.0 / self.world)
        else:
            self.opt.step(grad_scale=1.0 / self.world)
        return self._loss


# ------------------------------------------------------------------------------------ loops
def train_epoch(model, train_loader, criterion, optimizer, device, metrics=None, log_interval: int = 10):
    """Counterpart of train_epoch (/root/reference/src/train.py:68-127) in the reference's own call
    order (model -> criterion -> backward -> optimizer.step); fp32 throughout (no GradScaler)."""
    from .metrics import logits_metrics
    model.train()
    total = torch.zeros((), device=device)
    nb = 0
    mvals = {k: 0.0 for k in ("iou", "f1", "accuracy", "recall", "precision")}
    interval = max(1, len(train_loader) // 10)
    for i, (images, masks) in enumerate(train_loader):
        images = images.to(device, non_blocking=True)
        masks = masks.to(device, non_blocking=True)
        optimizer.zero_grad()
        outputs = model(images)
        if masks.dim() == 3:
            masks = masks.unsqueeze(1)
        loss = criterion(outputs, masks)
        loss.backward()
        optimizer.step()
        total += loss.detach()
        nb += 1
        if i % interval == 0:
            for k, v in logits_metrics(outputs.detach(), masks).items():
                mvals[k] += v
    nb = max(nb, 1)
    ncalc = max(1, (nb + interval - 1) // interval)
    return float(total) / nb, {k: v / ncalc for k, v in mvals.items()}


@torch.no_grad()
def validate(model, val_loader, criterion, device):
    """Counterpart of validate (/root/reference/src/train.py:129-173)."""
    from .metrics import logits_metrics
    model.eval()
    total = torch.zeros((), device=device)
    nb = 0
    mvals = {k: 0.0 for k in ("iou", "f1", "accuracy", "recall", "precision")}
    for images, masks in val_loader:
        images = images.to(device, non_blocking=True)
        masks = masks.to(device, non_blocking=True)
        outputs = model(images)
        if masks.dim() == 3:
            masks = masks.unsqueeze(1)
        total += criterion(outputs, masks)
        nb += 1
        for k, v in logits_metrics(outputs, masks).items():
            mvals[k] += v
    nb = max(nb, 1)
    return float(total) / nb, {k: v / nb for k, v in mvals.items()}
