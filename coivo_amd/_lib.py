"""ctypes binding of libcolvo.so -- the C-ABI declared in include/colvo.h.

There is NO fallback: if the HIP library is missing or a call fails, this raises.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# COLVO_LIB_PATH: developer override (ablation / instrumented builds made by tools/ablate_conv.sh)
LIB_PATH = os.environ.get("COLVO_LIB_PATH") or os.path.join(_HERE, "lib", "libcolvo.so")

F32, BF16 = 0, 1
ABI_VERSION = 10

_vp, _i, _f, _sz = C.c_void_p, C.c_int, C.c_float, C.c_size_t


class ConvDesc(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        "dtype", "B", "Ho", "Wo", "Cout", "ksize", "stride", "relu",
        "C0", "up0", "C1", "up1", "Hi", "Wi")]


class AdamArena(C.Structure):
    """Mirror of ColvoAdamArena (include/colvo.h)."""
    _fields_ = [("param", C.c_void_p), ("grad", C.c_void_p), ("exp_avg", C.c_void_p), ("exp_avg_sq", C.c_void_p), ("n", C.c_size_t)]


class WgradSlabs(C.Structure):
    """Mirror of ColvoWgradSlabs (include/colvo.h)."""
    _fields_ = [("scratch", C.c_void_p), ("dw", C.c_void_p), ("db", C.c_void_p), ("nsplit", C.c_int32), ("Cout", C.c_int32),
                ("Ctot", C.c_int32), ("pad_", C.c_int32)]


WGRAD_GROUP_MAX = 16
MAX_ARENAS = 4
ADAM_PLAIN_PER_WG = 2048          # COLVO_ADAM_PLAIN_PER_WG


# name -> (restype, argtypes); lists every symbol include/colvo.h declares
class Cmd(C.Structure):
    """Mirror of ColvoCmd (include/colvo.h)."""
    _fields_ = [("op", C.c_int32), ("stream", C.c_int32), ("desc", ConvDesc), ("p", C.c_void_p * 12),
                ("i", C.c_int32 * 12), ("f", C.c_float * 4)]


(CMD_CONV_FWD, CMD_CONV_DGRAD, CMD_CONV_WGRAD, CMD_PACK_NCHW, CMD_UNPACK_NHWC_GRAD, CMD_DEPTH_HEAD_FWD,
 CMD_DEPTH_HEAD_BWD, CMD_DEPTH_HEAD_WGRAD, CMD_POSE_HEAD_FWD, CMD_POSE_HEAD_BWD, CMD_FORK, CMD_JOIN,
 CMD_DEPTH_HEAD_BWD_PARTS, CMD_CONV_DGRAD_BOTH, CMD_WGRAD_REDUCE_GROUP, CMD_CONV_DGRAD_PLANES, CMD_CONV_BWD_FUSED,
 CMD_HEAD_WGRAD_REDUCE, CMD_CONV_HEAD_FUSED, CMD_PACK_STEM_POSE, CMD_HEAD_WGRAD_MFMA, CMD_SIDE_SYNC) = range(1, 23)
NPTR = 12      # pointer slots of a ColvoCmd

SIGNATURES = {
    "colvo_abi_version": (_i, []),
    "colvo_last_error": (C.c_char_p, []),
    "colvo_warp_loss_workspace_floats": (_sz, [_i, _i, _i]),
    "colvo_warp_loss_fwd": (_i, [_vp] * 7 + [_i, _i, _i, _f, _vp, _vp, _vp]),
    "colvo_warp_loss_bwd": (_i, [_vp] * 7 + [_i, _i, _i, _f] + [_vp] * 8),
    "colvo_warp_loss_fused": (_i, [_vp] * 7 + [_i, _i, _i, _f] + [_vp] * 6),
    "colvo_warp_loss_fused_bwd": (_i, [_vp] * 5 + [_i, _i, _i] + [_vp] * 5),
    "colvo_warp_loss_fused_bwd_params": (_i, [_vp] * 4 + [_i] + [_vp] * 4),
    "colvo_warp_loss_rescale": (_i, [_vp, _i, _vp]),
    "colvo_warp_loss_rescale_to": (_i, [_vp, _i, _vp, _vp]),
    "colvo_inverse_warp": (_i, [_vp] * 4 + [_i] * 4 + [_vp] * 3),
    "colvo_geo_loss_workspace_floats": (_sz, [_i, _i, _i]),
    "colvo_geo_loss_fwd": (_i, [_vp] * 4 + [_i, _i, _i] + [_vp] * 3),
    "colvo_geo_loss_bwd": (_i, [_vp] * 4 + [_i, _i, _i] + [_vp] * 7),
    "colvo_smooth_loss_fwd": (_i, [_vp, _vp, _i, _i, _i, _vp, _vp, _vp]),
    "colvo_smooth_loss_bwd": (_i, [_vp, _vp, _i, _i, _i, _vp, _vp, _vp]),
    "colvo_avgpool2_fwd": (_i, [_vp, _i, _i, _i, _vp, _vp]),
    "colvo_avgpool2_bwd": (_i, [_vp, _i, _i, _i, _vp, _vp]),
    "colvo_adam_step_multi": (_i, [_vp, _i, _f, _f, _f, _f, _f, _i, _vp]),
    "colvo_zero_multi": (_i, [_vp, _vp, _i, _vp]),
    "colvo_adam_pack_step": (_i, [_i, _vp, _i, _i, _f, _f, _f, _f, _f, _vp, _i, _vp]),
    "colvo_adam_pack_step_scaled": (_i, [_i, _vp, _i, _i, _f, _f, _f, _f, _f, _vp, _vp, _i, _vp]),
    "colvo_full_objective_workspace_floats": (_sz, [_i, _i, _i, _i]),
    "colvo_full_objective_fwd": (_i, [_vp] * 8 + [_i] * 4 + [_f] * 3 + [_vp] * 3),
    "colvo_full_objective_bwd": (_i, [_vp] * 3 + [_i] * 4 + [_f, _f] + [_vp] * 6),
    "colvo_full_objective_terms": (_i, [_vp, _i, _i, _i, _i, C.POINTER(_vp)]),
    "colvo_conv_fwd": (_i, [C.POINTER(ConvDesc)] + [_vp] * 6),
    "colvo_conv_dgrad": (_i, [C.POINTER(ConvDesc), _i, _vp, _vp, _vp, _vp, _i, _vp]),
    "colvo_conv_dgrad_both": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "colvo_conv_dgrad_planes": (_i, [C.POINTER(ConvDesc), _vp, _vp, _i, _i, _vp, _i, _vp]),
    "colvo_conv_head_fused_ok": (_i, [C.POINTER(ConvDesc)]),
    "colvo_conv_head_fused": (_i, [C.POINTER(ConvDesc), _vp, _vp, _vp, _vp, _vp, _f, _f, _vp, _vp, _vp, _vp]),
    "colvo_pack_stem_pose": (_i, [_vp, _i, _i, _i, _vp, _vp, _vp]),
    "colvo_conv_bwd_fused_ok": (_i, [C.POINTER(ConvDesc)]),
    "colvo_conv_bwd_fused": (_i, [C.POINTER(ConvDesc), _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "colvo_conv_bwd_fused_head_rows": (_i, [C.POINTER(ConvDesc)]),
    "colvo_depth_head_wgrad_reduce": (_i, [_vp, _i, _vp, _vp, _vp]),
    "colvo_depth_head_wgrad_mfma_rows": (_i, [_i, _i, _i]),
    "colvo_depth_head_wgrad_mfma": (_i, [_vp, _vp, _i, _i, _i, _vp, _vp]),
    "colvo_conv_wgrad": (_i, [C.POINTER(ConvDesc)] + [_vp] * 6),
    "colvo_conv_wgrad_clean": (_i, [_vp] * 6 + [_i, _vp]),
    "colvo_conv_wgrad_scratch_bytes": (_sz, [C.POINTER(ConvDesc)]),
    "colvo_conv_wgrad_det": (_i, [C.POINTER(ConvDesc)] + [_vp] * 6 + [_sz, _vp]),
    "colvo_conv_wgrad_splits": (_i, [C.POINTER(ConvDesc)]),
    "colvo_conv_wgrad_slabs": (_i, [C.POINTER(ConvDesc)] + [_vp] * 4 + [_sz, _vp]),
    "colvo_wgrad_reduce_group": (_i, [_vp, _i, _vp]),
    "colvo_depth_head_wgrad_scratch_bytes": (_sz, [_i, _i, _i, _i]),
    "colvo_depth_head_wgrad_det": (_i, [_i, _vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp, _sz, _vp]),
    "colvo_pose_head_bwd_det": (_i, [_i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _f, _f, _vp, _vp, _vp, _vp]),
    "colvo_relu_bwd_inplace": (_i, [_i, _vp, _vp, _sz, _vp]),
    "colvo_pack_weights": (_i, [_i, _vp, _i, _i, _i, _vp, _vp, _vp]),
    "colvo_pack_weights_multi": (_i, [_i, _vp, _vp, _i, _i, _vp, _vp, _vp]),
    "colvo_pack_nchw": (_i, [_i, C.POINTER(_vp), C.POINTER(C.c_int32), _i, _i, _i, _i, _i, _vp, _vp]),
    "colvo_unpack_nhwc_grad": (_i, [_i, _vp, _i, _i, _i, _i, _i, _i, _vp, _i, _vp]),
    "colvo_depth_head_fwd": (_i, [_i, _vp, _vp, _vp, _i, _i, _i, _i, _f, _f, _vp, _vp]),
    "colvo_depth_head_bwd": (_i, [_i, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _f, _f, _vp, _vp, _vp, _vp, _vp]),
    "colvo_depth_head_wgrad": (_i, [_i, _vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp]),
    "colvo_depth_head_bwd_parts": (_i, [_i] + [_vp] * 9 + [_i, _i, _i, _i, _f, _f] + [_vp] * 5),
    "colvo_pose_head_fwd": (_i, [_i, _vp, _vp, _vp, _i, _i, _i, _f, _f, _vp, _vp]),
    "colvo_pose_head_bwd": (_i, [_i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _f, _f, _vp, _vp, _vp, _vp]),
    "colvo_adam_step": (_i, [_vp, _vp, _vp, _vp, _sz, _f, _f, _f, _f, _f, _vp, _vp]),
    "colvo_adam_step_t": (_i, [_vp, _vp, _vp, _vp, _sz, _f, _f, _f, _f, _f, _i, _vp]),
    "colvo_cast_f32_bf16": (_i, [_vp, _vp, _sz, _i, _vp]),
    "colvo_zero": (_i, [_vp, _sz, _vp]),
    "colvo_frames_u8_to_f32": (_i, [_vp, _i, _i, _i, _i, _i, _vp, _vp]),
    "colvo_read_npy_u8_frames": (_i, [_vp, _i, _i, _i, _vp, _i]),
    "colvo_backproject": (_i, [_vp, _vp, _vp, _i, _i, _i, _vp, _vp]),
    "colvo_stitch_workspace_ints": (_sz, [_i, _i, _i, _i]),
    "colvo_stitch_point_cloud": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _f, _vp, _vp, _vp, _vp]),
    "colvo_run_commands": (_i, [_vp, _i, _vp, _vp]),
    "colvo_set_aux_side_streams": (_i, [_i]),
    "colvo_set_capture_policy": (_i, [_i, _i]),
    "colvo_set_capture_carry": (_i, [_i]),
    "colvo_capture_join": (_i, [_vp]),
    "colvo_capture_reset": (_i, [_vp]),
    "colvo_graph_stats": (_i, [_vp, C.POINTER(C.c_longlong), _i]),
    "colvo_graph_stats_reset": (_i, []),
    "colvo_form_counts": (_i, [C.POINTER(C.c_longlong), _i]),
    "colvo_form_counts_reset": (None, []),
    "colvo_form_name": (C.c_char_p, [_i]),
    "colvo_tune_set": (_i, [C.c_char_p, C.c_double]),
    "colvo_tune_get": (_i, [C.c_char_p, C.POINTER(C.c_double)]),
}

_lib = None


def load() -> C.CDLL:
    """Load libcolvo.so (built by `python -m coivo_amd.build` / __graft_entry__.build())."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} not found: the HIP extension is not built (run `python -m coivo_amd.build`). "
            "coivo_amd has no CPU or PyTorch fallback.")
    if not os.environ.get("COLVO_LIB_PATH"):
        from . import build as _build
        if not _build.is_current():
            raise RuntimeError(
                f"{LIB_PATH} was not built from the sources in this tree (coivo_amd/csrc, include/colvo.h or the "
                "build flags changed): run `python -m coivo_amd.build` (tests, bench.py and smoke() call build.ensure())")
    # torch bundles its own HIP/HSA runtime (soname libamdhip64.so.7, same as /opt/rocm's).  It must be
    # in the process BEFORE libcolvo.so so that our NEEDED entry binds to that copy: the kernels then share
    # torch's HIP context and streams.  Loading /opt/rocm's runtime first puts two HSA runtimes in one
    # process and every launch fails with "no ROCm-capable device is detected".
    import torch  # noqa: F401
    lib = C.CDLL(LIB_PATH)
    missing = []
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError:
            missing.append(name)
            continue
        fn.restype = res
        fn.argtypes = args
    if missing:
        raise RuntimeError(f"{LIB_PATH} does not export {missing}: stale build? run `python -m coivo_amd.build --force`")
    v = lib.colvo_abi_version()
    if v != ABI_VERSION:
        raise RuntimeError(f"libcolvo ABI version {v} != expected {ABI_VERSION}")
    _lib = lib
    return lib


def check(rc: int, what: str) -> None:
    if rc != 0:
        msg = load().colvo_last_error().decode("utf-8", "replace")
        raise RuntimeError(f"{what} failed (code {rc}): {msg}")


def tune_set(name: str, value: float) -> None:
    """Developer / test hook: set an entry of the library's tuning table (coivo_amd/csrc/tuning.h)."""
    check(load().colvo_tune_set(name.encode(), float(value)), "colvo_tune_set")


def tune_get(name: str) -> float:
    v = C.c_double()
    check(load().colvo_tune_get(name.encode(), C.byref(v)), "colvo_tune_get")
    return v.value


def form_counts(reset: bool = False) -> dict:
    """{form name: launches since the last reset} of the kernel forms the library's dispatchers choose by grid size (include/colvo.h
    colvo_form_counts); reset=True clears the counters after reading."""
    lib = load()
    buf = (C.c_longlong * 32)()
    n = lib.colvo_form_counts(buf, 32)
    out = {lib.colvo_form_name(i).decode(): int(buf[i]) for i in range(n)}
    if reset:
        lib.colvo_form_counts_reset()
    return out


def dev_env(name: str, default=None):
    """Developer switches of the Python side are honoured only under COLVO_DEV=1 (production reads no tuning variable)."""
    if os.environ.get("COLVO_DEV", "0") not in ("", "0"):
        return os.environ.get(name, default)
    return default


def ptr(t) -> int:
    """Device pointer of a torch tensor (None -> NULL)."""
    return 0 if t is None else t.data_ptr()


def stream_ptr() -> int:
    """hipStream_t of torch's current stream on the current device (the raw C call: torch.cuda.current_stream() builds a
    Python Stream object through four layers of device-index helpers, ~10 us a call and nine calls a step)."""
    import torch
    return torch._C._cuda_getCurrentRawStream(torch._C._cuda_getDevice())
