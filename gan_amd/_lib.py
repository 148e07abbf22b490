"""ctypes binding of libgan_amd.so (the C ABI declared in include/gan_amd.h).

The product path has NO CPU fallback: importing this module without the built library, or calling any
op when the library is missing, raises.  Build with ``python -c "import __graft_entry__ as g; g.build()"``
or ``make -C gan_amd/csrc``.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("GAN_AMD_LIB") or os.path.join(_HERE, "libgan_amd.so")   # (GAN_AMD_LIB: tools/diag_build.sh variant)

F32, BF16, F16 = 0, 1, 2
ACT_NONE, ACT_LRELU, ACT_RELU, ACT_TANH = 0, 1, 2, 3
E_ARG, E_SHAPE, E_WORKSPACE = -1, -2, -3
ACTS = {None: ACT_NONE, 'none': ACT_NONE, 'lrelu': ACT_LRELU, 'relu': ACT_RELU, 'tanh': ACT_TANH}


class GanTensor(C.Structure):
    _fields_ = [("ptr", C.c_void_p), ("n", C.c_int32), ("h", C.c_int32), ("w", C.c_int32), ("c", C.c_int32),
                ("pitch", C.c_int32)]


class GanPrepEntry(C.Structure):
    _fields_ = [("master", C.c_void_p), ("nk_native", C.c_void_p), ("nk_transposed", C.c_void_p), ("A", C.c_int32),
                ("B", C.c_int32), ("tile_start", C.c_int32), ("tiles_b", C.c_int32)]


class _Desc(C.Structure):
    """Descriptor structs carry their own size first (include/gan_amd.h): filled in here, so call sites list only the
    real fields; the library rejects a size it was not built with (GAN_E_ARG)."""

    def __init__(self, *args, **kw):
        super().__init__(C.sizeof(type(self)), *args, **kw)


class GanBwdFuse(C.Structure):
    _fields_ = [("ref", GanTensor), ("add", GanTensor), ("mean", C.c_void_p), ("rstd", C.c_void_p), ("gamma", C.c_void_p),
                ("beta", C.c_void_p), ("dropmask", C.c_void_p), ("mask_pitch", C.c_int32), ("act", C.c_int32), ("slope", C.c_float),
                ("cols", C.c_int32)]


class GanNormFuse(C.Structure):
    _fields_ = [("out", GanTensor), ("gamma", C.c_void_p), ("beta", C.c_void_p), ("mean", C.c_void_p), ("rstd", C.c_void_p),
                ("moving_mean", C.c_void_p), ("moving_var", C.c_void_p), ("eps", C.c_float), ("momentum", C.c_float),
                ("dropmask", C.c_void_p), ("act", C.c_int32), ("slope", C.c_float), ("dgamma", C.c_void_p), ("dbeta", C.c_void_p),
                ("accumulate", C.c_int32)]


class GanConvDesc(_Desc):
    _fields_ = [("struct_size", C.c_uint32), ("dtype", C.c_int32), ("stride", C.c_int32), ("x", GanTensor), ("y", GanTensor), ("w", C.c_void_p),
                ("w_rows", C.c_int32), ("bias", C.c_void_p), ("act", C.c_int32), ("slope", C.c_float),
                ("y_f32", C.c_int32), ("workspace", C.c_void_p), ("workspace_bytes", C.c_size_t),
                ("stats_partial", C.c_void_p), ("stats_groups", C.c_int32), ("stats_partial_bytes", C.c_size_t), ("bwd_fuse", C.c_void_p),
                ("norm_fuse", C.c_void_p)]


class GanAdamFuse(C.Structure):
    _fields_ = [("master", C.c_void_p), ("m", C.c_void_p), ("v", C.c_void_p), ("nk_native", C.c_void_p), ("nk_transposed", C.c_void_p),
                ("lr_t", C.c_void_p), ("beta1", C.c_float), ("beta2", C.c_float), ("eps", C.c_float)]


class GanWgradDesc(_Desc):
    _fields_ = [("struct_size", C.c_uint32), ("dtype", C.c_int32), ("stride", C.c_int32), ("big", GanTensor), ("small", GanTensor),
                ("dw", C.c_void_p), ("big_c", C.c_int32), ("small_c", C.c_int32), ("accumulate", C.c_int32),
                ("workspace", C.c_void_p), ("workspace_bytes", C.c_size_t), ("concurrent", C.c_int32), ("adam_fuse", C.c_void_p),
                ("dw_wire", C.c_void_p)]


class GanNormDesc(_Desc):
    _fields_ = [("struct_size", C.c_uint32), ("dtype", C.c_int32), ("y", GanTensor), ("a", GanTensor), ("groups", C.c_int32), ("eps", C.c_float),
                ("gamma", C.c_void_p), ("beta", C.c_void_p), ("mean", C.c_void_p), ("rstd", C.c_void_p),
                ("moving_mean", C.c_void_p), ("moving_var", C.c_void_p), ("momentum", C.c_float),
                ("dropmask", C.c_void_p), ("act", C.c_int32), ("slope", C.c_float), ("workspace", C.c_void_p),
                ("workspace_bytes", C.c_size_t), ("sync", C.c_void_p)]


class GanNormBwdDesc(_Desc):
    _fields_ = [("struct_size", C.c_uint32), ("dtype", C.c_int32), ("y", GanTensor), ("da", GanTensor), ("da2", GanTensor), ("dy", GanTensor),
                ("groups", C.c_int32), ("gamma", C.c_void_p), ("beta", C.c_void_p), ("mean", C.c_void_p),
                ("rstd", C.c_void_p), ("dropmask", C.c_void_p), ("act", C.c_int32), ("slope", C.c_float),
                ("dgamma", C.c_void_p), ("dbeta", C.c_void_p), ("accumulate", C.c_int32), ("workspace", C.c_void_p),
                ("workspace_bytes", C.c_size_t), ("sync", C.c_void_p)]


class GanActBwdDesc(_Desc):
    _fields_ = [("struct_size", C.c_uint32), ("dtype", C.c_int32), ("a", GanTensor), ("da", GanTensor), ("da2", GanTensor), ("dy", GanTensor),
                ("act", C.c_int32), ("slope", C.c_float), ("dbias", C.c_void_p), ("accumulate", C.c_int32),
                ("workspace", C.c_void_p), ("workspace_bytes", C.c_size_t)]


# name -> (restype, argtypes); every symbol include/gan_amd.h declares
SYMBOLS = {
    "gan_conv2d_fwd": (C.c_int, [C.POINTER(GanConvDesc), C.c_void_p]),
    "gan_conv2d_dgrad": (C.c_int, [C.POINTER(GanConvDesc), C.c_void_p]),
    "gan_convT2d_fwd": (C.c_int, [C.POINTER(GanConvDesc), C.c_void_p]),
    "gan_convT2d_dgrad": (C.c_int, [C.POINTER(GanConvDesc), C.c_void_p]),
    "gan_conv_workspace_bytes": (C.c_size_t, [C.POINTER(GanConvDesc), C.c_int]),
    "gan_conv_plan_info": (C.c_int, [C.POINTER(GanConvDesc), C.c_int, C.POINTER(C.c_int32)]),
    "gan_wgrad_plan_info": (C.c_int, [C.POINTER(GanWgradDesc), C.POINTER(C.c_int32)]),
    "gan_conv_wgrad": (C.c_int, [C.POINTER(GanWgradDesc), C.c_void_p]),
    "gan_wgrad_workspace_bytes": (C.c_size_t, [C.POINTER(GanWgradDesc)]),
    "gan_conv_stack_eligible": (C.c_int, [C.POINTER(GanConvDesc), C.c_int]),
    "gan_conv_tap_shared": (C.c_int, [C.POINTER(GanConvDesc), C.c_int]),
    "gan_conv_stack_plan_bytes": (C.c_size_t, [C.c_int32]),
    "gan_conv_stack_plan": (C.c_int, [C.c_void_p, C.POINTER(C.c_int32), C.c_int32, C.c_void_p, C.c_size_t]),
    "gan_conv_stack_launch": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "gan_conv_stack_barrier_bytes": (C.c_size_t, []),
    "gan_wgrad_adam_fused": (C.c_int, [C.POINTER(GanWgradDesc)]),
    "gan_wgrad_wire_direct": (C.c_int, [C.POINTER(GanWgradDesc)]),
    "gan_weights_prepare": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]),
    "gan_weights_prepare_multi": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]),
    "gan_adam_prepare_multi": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p,
                                         C.c_void_p, C.c_void_p, C.c_float, C.c_float, C.c_float, C.c_float, C.c_void_p, C.c_int32, C.c_void_p]),
    "gan_norm_stats": (C.c_int, [C.POINTER(GanNormDesc), C.c_void_p]),
    "gan_norm_stats_finalize": (C.c_int, [C.POINTER(GanNormDesc), C.c_int32, C.c_void_p]),
    "gan_norm_act_fwd": (C.c_int, [C.POINTER(GanNormDesc), C.c_void_p]),
    "gan_norm_workspace_bytes": (C.c_size_t, [C.c_int32, C.c_int32, C.c_int64]),
    "gan_norm_stats_partial": (C.c_int, [C.POINTER(GanNormDesc), C.c_void_p]),
    "gan_norm_finalize_act_fwd": (C.c_int, [C.POINTER(GanNormDesc), C.c_int32, C.c_void_p]),
    "gan_norm_sync_bytes": (C.c_size_t, []),
    "gan_norm_sync_error_offset": (C.c_size_t, []),
    "gan_norm_act_bwd": (C.c_int, [C.POINTER(GanNormBwdDesc), C.c_void_p]),
    "gan_norm_act_bwd_fused": (C.c_int, [C.POINTER(GanNormBwdDesc), C.c_int32, C.c_void_p]),
    "gan_act_bwd": (C.c_int, [C.POINTER(GanActBwdDesc), C.c_void_p]),
    "gan_bce_logits": (C.c_int, [C.c_void_p, C.c_int64, C.c_float, C.c_float, C.c_int32, C.c_void_p, C.c_float,
                                 C.c_int32, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]),
    "gan_patchgan_losses": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32,
                                      C.c_float, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "gan_l1": (C.c_int, [C.c_int32, C.POINTER(GanTensor), C.POINTER(GanTensor), C.c_float, C.c_int32, C.c_void_p,
                         C.c_float, C.POINTER(GanTensor), C.c_void_p, C.c_void_p, C.c_void_p]),
    "gan_adam_begin": (C.c_int, [C.c_void_p, C.c_void_p, C.c_float, C.c_float, C.c_float, C.c_void_p, C.c_void_p]),
    "gan_adam_tf": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_float,
                              C.c_float, C.c_float, C.c_float, C.c_void_p, C.c_int32, C.c_void_p]),
    "gan_sum3": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p]),
    "gan_grads_check": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]),
    "gan_loss_scale_update": (C.c_int, [C.c_void_p, C.c_int32, C.c_float, C.c_void_p]),
    "gan_dropout_mask": (C.c_int, [C.c_void_p, C.c_int64, C.c_uint64, C.c_void_p, C.c_uint32, C.c_void_p]),
    "gan_pack": (C.c_int, [C.c_int32, C.c_void_p, C.POINTER(GanTensor), C.c_void_p]),
    "gan_pack_multi": (C.c_int, [C.c_int32, C.c_int32, C.POINTER(C.c_void_p), C.POINTER(GanTensor), C.c_void_p]),
    "gan_dropout_mask_multi": (C.c_int, [C.c_int32, C.POINTER(C.c_void_p), C.POINTER(C.c_int64), C.c_uint64, C.c_void_p,
                                         C.POINTER(C.c_uint32), C.c_void_p, C.c_void_p]),
    "gan_unpack": (C.c_int, [C.c_int32, C.POINTER(GanTensor), C.c_void_p, C.c_void_p]),
    "gan_copy_view": (C.c_int, [C.c_int32, C.POINTER(GanTensor), C.POINTER(GanTensor), C.c_void_p]),
    "gan_bias_grad": (C.c_int, [C.c_int32, C.POINTER(GanTensor), C.c_void_p, C.c_int32, C.c_void_p, C.c_size_t, C.c_void_p]),
    "gan_grad_pack": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    "gan_grad_unpack": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_float, C.c_void_p]),
    "gan_crc32c": (C.c_uint32, [C.c_uint32, C.c_void_p, C.c_size_t]),
    "gan_version": (C.c_char_p, []),
    "gan_set_option": (C.c_int, [C.c_char_p, C.c_int32]),
    "gan_get_option": (C.c_int, [C.c_char_p, C.POINTER(C.c_int32)]),
    "gan_launch_log": (C.c_size_t, [C.c_char_p, C.c_size_t]),
}

_lib = None


class GanAmdError(RuntimeError):
    pass


def load():
    """Load the shared library (once) and bind every declared symbol; raises if anything is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise GanAmdError(f"{LIB_PATH} not built: the MI355X HIP extension is required (no CPU fallback). "
                          "Run `make -C gan_amd/csrc` or __graft_entry__.build().")
    # PyTorch-ROCm ships its own libamdhip64: if this library were loaded first it would bind /opt/rocm's copy, torch would then load its
    # own, and streams / pointers created by one runtime are strangers to the other (every launch: hipErrorNoDevice).  Loading torch
    # first makes both resolve to ONE runtime.  (No torch in the process, e.g. the plain C caller: nothing to order.)
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)       # AttributeError if the library does not export it
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def set_option(key, value):
    """Planner option of the library (include/gan_amd.h, gan_set_option); returns the previous value."""
    lib = load()
    old = C.c_int32()
    check(lib.gan_get_option(key.encode(), C.byref(old)), f"gan_get_option({key})")
    check(lib.gan_set_option(key.encode(), int(value)), f"gan_set_option({key})")
    return old.value


def get_option(key):
    lib = load()
    v = C.c_int32()
    check(lib.gan_get_option(key.encode(), C.byref(v)), f"gan_get_option({key})")
    return v.value


def launch_log():
    """Kernel symbols (mangled) of every launch recorded since set_option('diag.launch_log', 1), in enqueue order."""
    lib = load()
    need = lib.gan_launch_log(None, 0)
    buf = C.create_string_buffer(need)
    lib.gan_launch_log(buf, need)
    return [ln for ln in buf.value.decode().split('\n') if ln]


def check(rc, what):
    if rc != 0:
        kind = {-1: "invalid argument", -2: "unsupported shape", -3: "workspace too small"}.get(rc, f"hipError {rc}")
        raise GanAmdError(f"{what} failed: {kind}")
