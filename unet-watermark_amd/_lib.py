"""ctypes binding of libuwm.so (include/uwm.h) — the stub INTEGRATION.md describes.

The product path has NO fallback: if the HIP library is missing or fails to load, every entry
point raises.  Nothing here imports `oracle/`.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
import sys
from pathlib import Path

# more hardware queues than ROCm's default 4, so the library's side stream does not alias the compute stream's queue
# once RCCL / other libraries have created their own streams (effective only if the HIP runtime is not up yet)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

_PKG_DIR = Path(__file__).resolve().parent
_CSRC = _PKG_DIR / "csrc"
LIB_PATH = _PKG_DIR / "libuwm.so"
SOURCES = ["conv_igemm.hip", "conv_patch.hip", "conv_patch16.hip", "conv_wino.hip", "conv_wino8.hip", "conv_wino_x3.hip", "conv_f16x3.hip", "conv_f16x3v2.hip", "conv_stem_f16x3.hip", "conv_up2.hip", "conv_up2_f16.hip", "conv_c16_f16.hip", "conv_gemm.hip", "conv_head.hip", "mbconv.hip", "wgrad_igemm.hip", "wgrad_patch.hip", "wgrad_wino.hip", "wgrad_f16x3.hip", "wgrad_c16.hip", "wgrad_gemm.hip", "wgrad_stem.hip", "elementwise.hip", "loss.hip", "uwm_model.hip"]
HIP_ARCH = "gfx950"
# per-file compiler flags.  The fp16x3 kernels interleave their staging VALU work with MFMAs; hipcc turns the f4 arithmetic of that
# work into packed v_pk_{fma,mul,add}_f32, which cost more than two plain VALU instructions beside MFMAs (MI355X_MICROARCH.md,
# "packed f32 VALU ... an anti-lever beside MFMAs").  Without them: 1099 -> 1110 img/s on the headline step (same box, alternating
# runs); the exact-fp32 kernels measured the other way (782 vs 779) and keep the default.  (The host pass of hipcc ignores the
# feature with a warning.)
EXTRA_FLAGS = {name: ["-Xclang", "-target-feature", "-Xclang", "-packed-fp32-ops"]
               for name in ("conv_f16x3.hip", "conv_f16x3v2.hip", "wgrad_f16x3.hip", "conv_stem_f16x3.hip")}


class uwm_unet_desc(C.Structure):
    _fields_ = [("encoder", C.c_int), ("in_channels", C.c_int), ("classes", C.c_int),
                ("decoder_channels", C.c_int * 5), ("bn_eps", C.c_float), ("bn_momentum", C.c_float), ("arch", C.c_int)]


class uwm_tensor_info(C.Structure):
    _fields_ = [("name", C.c_char * 96), ("kind", C.c_int), ("arena", C.c_int), ("ndim", C.c_int),
                ("offset", C.c_longlong), ("shape", C.c_longlong * 4), ("stride", C.c_longlong * 4)]


class uwm_src(C.Structure):
    _fields_ = [("ptr", C.c_void_p), ("scale", C.c_void_p), ("shift", C.c_void_p),
                ("C", C.c_int), ("H", C.c_int), ("W", C.c_int), ("up", C.c_int), ("relu", C.c_int)]


KIND_CONV_W, KIND_BIAS, KIND_BN_GAMMA, KIND_BN_BETA, KIND_BN_MEAN, KIND_BN_VAR = range(6)
ARENA_PARAM, ARENA_BUFFER = 0, 1
ENC = {"resnet18": 18, "resnet34": 34, "resnet50": 50, "efficientnet-b4": 104}
ARCH = {"Unet": 0, "UnetPlusPlus": 1}
PREC = {"f32": 0, "bf16x3": 1, "bf16x3_all": 2, "f16x3": 3, "f16x3_all": 4, "f16x1": 5, "f16x3_bwd2": 6}
P, I, L, F, Z = C.c_void_p, C.c_int, C.c_longlong, C.c_float, C.c_size_t

# every symbol include/uwm.h declares: (restype, argtypes)
SIGNATURES = {
    "uwm_last_error": (C.c_char_p, []),
    "uwm_version": (I, []),
    "uwm_create": (I, [C.POINTER(uwm_unet_desc), C.POINTER(P)]),
    "uwm_destroy": (None, [P]),
    "uwm_param_arena_floats": (L, [P]),
    "uwm_buffer_arena_floats": (L, [P]),
    "uwm_param_count": (L, [P]),
    "uwm_num_tensors": (I, [P]),
    "uwm_tensor_info_get": (I, [P, I, C.POINTER(uwm_tensor_info)]),
    "uwm_logits_channels": (I, [P]),
    "uwm_num_stages": (I, [P]),
    "uwm_stage_range": (I, [P, I, C.POINTER(L), C.POINTER(L)]),
    "uwm_bind": (I, [P, P, P, P]),
    "uwm_workspace_bytes": (Z, [P, I, I, I, I]),
    "uwm_conv_flops": (I, [P, I, I, C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "uwm_forward": (I, [P, P, P, P, Z, I, I, I, I, P]),
    "uwm_backward": (I, [P, P, P, I, I, P]),
    "uwm_loss": (I, [P, I, P, I, L, F, F, F, F, P, P, P, I, F, P]),
    "uwm_loss_sums": (I, [P, I, P, I, L, P, P]),
    "uwm_loss_apply": (I, [P, I, P, I, L, L, F, F, F, F, P, P, P, I, F, P]),
    "uwm_stats": (I, [P, I, P, I, I, L, F, I, P, P]),
    "uwm_threshold": (I, [P, I, L, F, I, P, P]),
    "uwm_adam": (I, [P, P, P, P, L, F, F, F, F, F, L, F, P]),
    "uwm_adam_clip": (I, [P, P, P, P, L, F, F, F, F, F, L, F, F, P, P]),
    "uwm_adam_graph": (I, [P, P, P, P, L, P, P, P]),
    "uwm_sgd": (I, [P, P, P, L, F, F, F, L, F, F, P, P]),
    "uwm_scale": (I, [P, L, F, P]),
    "uwm_set_winograd": (I, [I]),
    "uwm_set_winograd_mode": (I, [P, I]),
    "uwm_get_winograd_mode": (I, [P]),
    "uwm_set_precision": (I, [P, I]),
    "uwm_get_precision": (I, [P]),
    "uwm_set_precision_fill": (I, [P, I]),
    "uwm_set_routing_batch": (I, [P, I]),
    "uwm_routing_enable": (I, [P, I]),
    "uwm_routing_dump": (L, [P, C.c_char_p, L, I]),
    "uwm_allreduce_grads": (I, [P, P, I, I, P]),
    "uwm_grad_arena": (P, [P]),
    "uwm_set_join_stream": (I, [P, P]),
    "uwm_set_drop_connect": (I, [P, P]),
    "uwm_op_depthwise": (I, [I, P, P, I, I, I, I, I, I, I, I, I, P, P, P, P]),
    "uwm_op_depthwise_scratch_floats": (L, [I, I, I, I, I]),
    "uwm_num_mbconv_blocks": (I, [P]),
    "uwm_mbconv_drop_rate": (F, [P, I]),
    "uwm_preprocess_u8": (I, [P, I, I, I, I, C.POINTER(C.c_float), C.POINTER(C.c_float), P, P, P]),
    "uwm_preprocess_mask_u8": (I, [P, I, I, I, I, P, P, P]),
    "uwm_resize_threshold": (I, [P, I, I, I, I, I, I, F, I, P, P, P]),
    "uwm_set_side_stream": (I, [P, I]),
    "uwm_prof_enable": (I, [I]),
    "uwm_prof_collect": (I, [C.POINTER(C.c_double), I]),
    "uwm_prof_class_name": (C.c_char_p, [I]),
    "uwm_debug_lookup": (I, [P, C.c_char_p, C.POINTER(L), C.POINTER(L)]),
    "uwm_op_set_igemm_f16x3": (I, [I]),
    "uwm_op_conv": (I, [C.POINTER(uwm_src), C.POINTER(uwm_src), P, I, I, I, I, I, I, I, I, P, P, P, I, P]),
    "uwm_op_dgrad": (I, [P, I, I, I, I, P, I, I, I, I, I, I, I, I, P, P, P, P, P, P]),
    "uwm_op_wgrad": (I, [C.POINTER(uwm_src), C.POINTER(uwm_src), P, I, I, I, I, I, I, I, I, I, I, P, I, P]),
    "uwm_op_pack_dgrad": (I, [P, I, I, I, I, P, I, I, P]),
    "uwm_op_maxpool": (I, [C.POINTER(uwm_src), I, P, P, P]),
    "uwm_op_maxpool_backward": (I, [P, P, P, C.POINTER(uwm_src), I, P, P]),
    "uwm_op_bn_backward": (I, [P, P, P, P, P, P, P, P, P, L, I, P]),
    "uwm_op_dgrad_upsplit": (I, [P, I, I, I, I, P, I, I, I, P, P, P, P, P, P]),
    "uwm_op_upsplit": (I, [P, I, I, I, I, I, P, P, P, P, P, P]),
    "uwm_op_residual": (I, [P, P, P, P, P, P, P, L, I, P]),
}

_lib = None


def build_library(force: bool = False, verbose: bool = False) -> Path:
    """Compile csrc/*.hip for gfx950 into libuwm.so (hipcc cross-compiles without a GPU)."""
    srcs = [_CSRC / s for s in SOURCES]
    hdrs = [_CSRC / "uwm_kernels.h", _PKG_DIR.parent / "include" / "uwm.h"] + sorted(_CSRC.glob("*.inc"))
    if LIB_PATH.exists() and not force:
        newest = max(p.stat().st_mtime for p in srcs + hdrs)
        if LIB_PATH.stat().st_mtime >= newest:
            return LIB_PATH
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objdir = _PKG_DIR / "build"
    objdir.mkdir(exist_ok=True)
    procs = []
    hdr_m = max(p.stat().st_mtime for p in hdrs + [Path(__file__)])
    for s in srcs:
        o = objdir / (s.stem + ".o")
        if not force and o.exists() and o.stat().st_mtime >= max(s.stat().st_mtime, hdr_m):
            procs.append((None, None, o))
            continue
        cmd = [hipcc, f"--offload-arch={HIP_ARCH}", "-O3", "-std=c++17", "-fPIC"] + EXTRA_FLAGS.get(s.name, []) + ["-c", str(s), "-o", str(o)]
        procs.append((cmd, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT), o))
    objs = []
    for cmd, p, o in procs:
        if p is None:
            objs.append(str(o))
            continue
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError(f"hipcc failed: {' '.join(cmd)}\n{out.decode(errors='replace')}")
        if verbose and out:
            sys.stderr.write(out.decode(errors="replace"))
        objs.append(str(o))
    cmd = [hipcc, f"--offload-arch={HIP_ARCH}", "-shared", "-fPIC", "-o", str(LIB_PATH)] + objs
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    if r.returncode != 0:
        raise RuntimeError(f"link failed: {' '.join(cmd)}\n{r.stdout.decode(errors='replace')}")
    return LIB_PATH


def lib():
    """Load libuwm.so (raises RuntimeError when the HIP extension is not built / not loadable)."""
    global _lib
    if _lib is not None:
        return _lib
    global LIB_PATH
    if os.environ.get("UWM_LIB"):          # experiments: load an alternative build of the same library
        LIB_PATH = Path(os.environ["UWM_LIB"])
    if not LIB_PATH.exists():
        raise RuntimeError(
            f"{LIB_PATH} not found: the HIP extension is not built. Run `python -c 'import __graft_entry__ as g; "
            f"g.build()'` (needs hipcc). There is no CPU fallback for this path.")
    try:
        h = C.CDLL(str(LIB_PATH))
    except OSError as e:  # missing libamdhip64 etc.
        raise RuntimeError(f"cannot load {LIB_PATH}: {e}") from e
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(h, name)       # AttributeError if a declared symbol is not exported
        fn.restype = res
        fn.argtypes = args
    _lib = h
    return _lib


def check(rc: int, exc=RuntimeError):
    if rc != 0:
        msg = lib().uwm_last_error().decode(errors="replace")
        raise exc(msg)


def stream_ptr(device=None) -> int:
    import torch
    return torch.cuda.current_stream(device).cuda_stream


class _NoGuard:
    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False


_NOGUARD = _NoGuard()


def on_device(t):
    """Context that makes `t`'s HIP device current for a launch through the free ABI functions (they enqueue on the
    caller's stream and expect its device to be current); free when it already is."""
    import torch
    if t.device.type == "cuda" and t.device.index is not None and t.device.index != torch.cuda.current_device():
        return torch.cuda.device(t.device)
    return _NOGUARD


_DT = None


def target_dtype_code(t) -> int:
    import torch
    global _DT
    if _DT is None:
        _DT = {torch.float32: 0, torch.int64: 1, torch.uint8: 2, torch.int32: 3, torch.bool: 2}
    if t.dtype not in _DT:
        raise TypeError(f"unsupported target dtype {t.dtype}")
    return _DT[t.dtype]
