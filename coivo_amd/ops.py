"""Op-level host wrappers over the C-ABI (include/colvo.h): tensors in, raw pointers + stream out.

Feature maps are NHWC torch tensors (float32 or bfloat16); everything is enqueued on PyTorch's current
stream.  No op here has a CPU or PyTorch fallback.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Sequence, Tuple

import torch

from . import _lib, program
from ._lib import ConvDesc

MIN_DEPTH, MAX_DEPTH = 0.1, 10.0
POSE_SCALE, LCC_SCALE = 0.01, 0.1


def dt_code(dtype: torch.dtype) -> int:
    if dtype == torch.float32:
        return _lib.F32
    if dtype == torch.bfloat16:
        return _lib.BF16
    raise TypeError(f"unsupported feature-map dtype {dtype}")


def _need_cuda(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise RuntimeError("coivo_amd ops run on the GPU only; there is no CPU fallback")


def conv_desc(dtype: torch.dtype, B: int, Hi: int, Wi: int, C0: int, Cout: int, *, stride: int = 1, relu: bool = True,
              C1: int = 0, up0: bool = False, up1: bool = False) -> ConvDesc:
    d = ConvDesc()
    d.dtype = dt_code(dtype)
    d.B, d.Hi, d.Wi = B, Hi, Wi
    d.Ho, d.Wo = (Hi - 1) // stride + 1, (Wi - 1) // stride + 1
    d.Cout, d.ksize, d.stride, d.relu = Cout, 3, stride, int(relu)
    d.C0, d.up0, d.C1, d.up1 = C0, int(up0), C1, int(up1)
    return d


def conv_fwd(d: ConvDesc, x0, x1, w_fwd, bias, y) -> None:
    _need_cuda(x0, x1, w_fwd, bias, y)
    rec = program.recording()
    if rec is not None:
        return rec.add(_lib.CMD_CONV_FWD, d, (x0, x1, w_fwd, bias, y))
    lib = _lib.load()
    _lib.check(lib.colvo_conv_fwd(C.byref(d), _lib.ptr(x0), _lib.ptr(x1), _lib.ptr(w_fwd), _lib.ptr(bias),
                                  _lib.ptr(y), _lib.stream_ptr()), "colvo_conv_fwd")


def conv_dgrad(d: ConvDesc, src: int, dy, w_bwd, relu_mask, dx, accumulate: bool) -> None:
    _need_cuda(dy, w_bwd, relu_mask, dx)
    rec = program.recording()
    if rec is not None:
        return rec.add(_lib.CMD_CONV_DGRAD, d, (dy, w_bwd, relu_mask, dx), (src, int(accumulate)))
    lib = _lib.load()
    _lib.check(lib.colvo_conv_dgrad(C.byref(d), src, _lib.ptr(dy), _lib.ptr(w_bwd), _lib.ptr(relu_mask),
                                    _lib.ptr(dx), int(accumulate), _lib.stream_ptr()), "colvo_conv_dgrad")


def conv_dgrad_both(d: ConvDesc, dy, w_bwd, relu_mask0, relu_mask1, dx0, dx1) -> None:
    """Input gradients w.r.t. both sources of a concat layer in one launch (colvo_conv_dgrad_both)."""
    _need_cuda(dy, w_bwd, relu_mask0, relu_mask1, dx0, dx1)
    rec = program.recording()
    if rec is not None:
        return rec.add(_lib.CMD_CONV_DGRAD_BOTH, d, (dy, w_bwd, relu_mask0, relu_mask1, dx0, dx1), ())
    lib = _lib.load()
    _lib.check(lib.colvo_conv_dgrad_both(C.byref(d), _lib.ptr(dy), _lib.ptr(w_bwd), _lib.ptr(relu_mask0), _lib.ptr(relu_mask1),
                                         _lib.ptr(dx0), _lib.ptr(dx1), _lib.stream_ptr()), "colvo_conv_dgrad_both")


def conv_dgrad_planes(d: ConvDesc, dy, w_master, c_begin: int, c_count: int, dst, accumulate: bool = False) -> None:
    """Input gradient w.r.t. channels [c_begin, c_begin + c_count) only, as fp32 planes dst [c_count, B, 1, Hi, Wi]
    (include/colvo.h colvo_conv_dgrad_planes)."""
    _need_cuda(dy, w_master, dst)
    rec = program.recording()
    if rec is not None:
        return rec.add(_lib.CMD_CONV_DGRAD_PLANES, d, (dy, w_master, dst), (c_begin, c_count, int(accumulate)))
    lib = _lib.load()
    _lib.check(lib.colvo_conv_dgrad_planes(C.byref(d), _lib.ptr(dy), _lib.ptr(w_master), c_begin, c_count, _lib.ptr(dst),
                                           int(accumulate), _lib.stream_ptr()), "colvo_conv_dgrad_planes")


def conv_head_fused_ok(d: ConvDesc) -> bool:
    return bool(_lib.load().colvo_conv_head_fused_ok(C.byref(d)))


def conv_head_fused(d: ConvDesc, x, w_fwd, bias, head_w, head_b, y, depth, pose_in=None) -> None:
    """The narrow full-resolution layer and the depth head behind it in one pass (include/colvo.h colvo_conv_head_fused): writes the
    layer's output y (NHWC bf16), depth [B,1,H,W] fp32 and -- with pose_in -- the two depth channels of PoseNet's input."""
    _need_cuda(x, w_fwd, bias, head_w, head_b, y, depth, pose_in)
    rec = program.recording()
    if rec is not None:
        return rec.add(_lib.CMD_CONV_HEAD_FUSED, d, (x, w_fwd, bias, head_w, head_b, y, depth, pose_in), (), (MIN_DEPTH, MAX_DEPTH))
    lib = _lib.load()
    _lib.check(lib.colvo_conv_head_fused(C.byref(d), _lib.ptr(x), _lib.ptr(w_fwd), _lib.ptr(bias), _lib.ptr(head_w), _lib.ptr(head_b),
                                         MIN_DEPTH, MAX_DEPTH, _lib.ptr(y), _lib.ptr(depth), _lib.ptr(pose_in), _lib.stream_ptr()),
               "colvo_conv_head_fused")


def pack_stem_pose(frames, stem, pose_in) -> None:
    """frames [2B,3,H,W] fp32 -> stem [2B,H,W,8] bf16 and the rgb channels of pose_in [B,H,W,8] bf16 (colvo_pack_stem_pose)."""
    _need_cuda(frames, stem, pose_in)
    B2, _, H, W = frames.shape
    rec = program.recording()
    if rec is not None:
        return rec.add(_lib.CMD_PACK_STEM_POSE, None, (frames, stem, pose_in), (B2, H, W))
    _lib.check(_lib.load().colvo_pack_stem_pose(_lib.ptr(frames), B2, H, W, _lib.ptr(stem), _lib.ptr(pose_in), _lib.stream_ptr()),
               "colvo_pack_stem_pose")


def conv_bwd_fused_ok(d: ConvDesc) -> bool:
    return bool(_lib.load().colvo_conv_bwd_fused_ok(C.byref(d)))


def conv_bwd_fused(d: ConvDesc, dy, w_bwd, x, relu_mask: bool, dx, dw, db, head_dpre=None, head_w=None, head_partials=None) -> None:
    """Input gradient (written to dx, masked by x > 0 when relu_mask) and weight / bias gradient (added to dw / db) of a qualifying
    narrow layer in one pass (include/colvo.h colvo_conv_bwd_fused).  head_dpre / head_w: the HEAD form -- `dy` is the layer's
    OUTPUT and the gradient is made on the fly from the depth head's d(pre) plane and weights; head_partials
    ([conv_bwd_fused_head_rows(d), 145] floats): the head's own weight gradient as partial rows for depth_head_wgrad_reduce."""
    _need_cuda(dy, w_bwd, x, dx, dw, db, head_dpre, head_w, head_partials)
    rec = program.recording()
    if rec is not None:
        return rec.add(_lib.CMD_CONV_BWD_FUSED, d, (dy, w_bwd, x, dx, dw, db, head_dpre, head_w, head_partials), (int(relu_mask),))
    lib = _lib.load()
    _lib.check(lib.colvo_conv_bwd_fused(C.byref(d), _lib.ptr(dy), _lib.ptr(w_bwd), _lib.ptr(x), int(relu_mask), _lib.ptr(dx),
                                        _lib.ptr(dw), _lib.ptr(db), _lib.ptr(head_dpre), _lib.ptr(head_w), _lib.ptr(head_partials),
                                        _lib.stream_ptr()), "colvo_conv_bwd_fused")


def conv_bwd_fused_head_rows(d: ConvDesc) -> int:
    return int(_lib.load().colvo_conv_bwd_fused_head_rows(C.byref(d)))


def depth_head_wgrad_mfma(y, dpre, dw, db) -> None:
    """The 16-channel bf16 depth head's weight / bias gradient by MFMA: partial rows + the table reduction (two launches on the current
    stream; include/colvo.h colvo_depth_head_wgrad_mfma)."""
    _need_cuda(y, dpre, dw, db)
    B, H, W, Cc = y.shape
    if Cc != 16 or y.dtype != torch.bfloat16:
        raise ValueError("depth_head_wgrad_mfma: the 16-channel bf16 head only")
    lib = _lib.load()
    rows = lib.colvo_depth_head_wgrad_mfma_rows(B, H, W)
    if rows <= 0:
        raise RuntimeError("colvo_depth_head_wgrad_mfma_rows failed")
    hp = torch.empty(rows * 145, device=y.device, dtype=torch.float32)
    rec = program.recording()
    if rec is not None:
        rec.add(_lib.CMD_HEAD_WGRAD_MFMA, None, (y, dpre, hp), (B, H, W))
        return rec.add(_lib.CMD_HEAD_WGRAD_REDUCE, None, (hp, dw, db), (int(rows),))
    _lib.check(lib.colvo_depth_head_wgrad_mfma(_lib.ptr(y), _lib.ptr(dpre), B, H, W, _lib.ptr(hp), _lib.stream_ptr()), "colvo_depth_head_wgrad_mfma")
    _lib.check(lib.colvo_depth_head_wgrad_reduce(_lib.ptr(hp), int(rows), _lib.ptr(dw), _lib.ptr(db), _lib.stream_ptr()),
               "colvo_depth_head_wgrad_reduce")


def depth_head_wgrad_reduce(partials, rows: int, dw, db) -> None:
    """dw [9][16] / db [1] += the column sums of partials [rows][145] (fixed order)."""
    _need_cuda(partials, dw, db)
    rec = program.recording()
    if rec is not None:
        return rec.add(_lib.CMD_HEAD_WGRAD_REDUCE, None, (partials, dw, db), (int(rows),))
    _lib.check(_lib.load().colvo_depth_head_wgrad_reduce(_lib.ptr(partials), int(rows), _lib.ptr(dw), _lib.ptr(db), _lib.stream_ptr()),
               "colvo_depth_head_wgrad_reduce")


def conv_wgrad_scratch(d: ConvDesc, device) -> torch.Tensor:
    """Scratch for the deterministic weight gradient of one conv call (include/colvo.h colvo_conv_wgrad_det)."""
    n = _lib.load().colvo_conv_wgrad_scratch_bytes(C.byref(d))
    if n == 0:
        raise RuntimeError("colvo_conv_wgrad_scratch_bytes failed: " + _lib.load().colvo_last_error().decode("utf-8", "replace"))
    return torch.empty((n + 3) // 4, device=device, dtype=torch.float32)


def conv_wgrad(d: ConvDesc, x0, x1, dy, dw, db, scratch: Optional[torch.Tensor] = None, arena_is_zero: bool = False) -> None:
    """dw / db += the layer's weight / bias gradient.  With `scratch` (conv_wgrad_scratch) the deterministic form: per-split slabs
    + a fixed-order second launch instead of float atomics.  arena_is_zero: the caller vouches that dw / db hold zeros -- single-split
    layers then store instead of adding (include/colvo.h colvo_conv_wgrad_clean)."""
    _need_cuda(x0, x1, dy, dw, db, scratch)
    nb = 0 if scratch is None else scratch.numel() * scratch.element_size()
    rec = program.recording()
    if rec is not None:
        # (i[2] = "the arena is still zero": patched per replay by Program.set_flags, nn._ArenaModule._run_pass)
        return rec.add(_lib.CMD_CONV_WGRAD, d, (x0, x1, dy, dw, db, scratch), (nb, 0, 0), flag_slot=2 if scratch is None else None)
    lib = _lib.load()
    if scratch is not None:
        _lib.check(lib.colvo_conv_wgrad_det(C.byref(d), _lib.ptr(x0), _lib.ptr(x1), _lib.ptr(dy), _lib.ptr(dw), _lib.ptr(db),
                                            _lib.ptr(scratch), nb, _lib.stream_ptr()), "colvo_conv_wgrad_det")
        return
    if arena_is_zero:
        _lib.check(lib.colvo_conv_wgrad_clean(C.byref(d), _lib.ptr(x0), _lib.ptr(x1), _lib.ptr(dy), _lib.ptr(dw), _lib.ptr(db), 1,
                                              _lib.stream_ptr()), "colvo_conv_wgrad_clean")
        return
    _lib.check(lib.colvo_conv_wgrad(C.byref(d), _lib.ptr(x0), _lib.ptr(x1), _lib.ptr(dy), _lib.ptr(dw),
                                    _lib.ptr(db), _lib.stream_ptr()), "colvo_conv_wgrad")


def conv_wgrad_splits(d: ConvDesc) -> int:
    """Number of pixel-range splits (= slabs) conv_wgrad_slabs / the deterministic form will write for this layer."""
    n = _lib.load().colvo_conv_wgrad_splits(C.byref(d))
    if n <= 0:
        raise RuntimeError("colvo_conv_wgrad_splits failed: " + _lib.load().colvo_last_error().decode("utf-8", "replace"))
    return n


def conv_wgrad_slabs(d: ConvDesc, x0, x1, dy, scratch: torch.Tensor) -> None:
    """First half of the grouped weight gradient (include/colvo.h colvo_conv_wgrad_slabs): every split stores its sums into its
    slab of `scratch` (conv_wgrad_scratch); nothing is added to dw / db until wgrad_reduce_group."""
    _need_cuda(x0, x1, dy, scratch)
    nb = scratch.numel() * scratch.element_size()
    rec = program.recording()
    if rec is not None:
        return rec.add(_lib.CMD_CONV_WGRAD, d, (x0, x1, dy, None, None, scratch), (nb, 1))
    lib = _lib.load()
    _lib.check(lib.colvo_conv_wgrad_slabs(C.byref(d), _lib.ptr(x0), _lib.ptr(x1), _lib.ptr(dy), _lib.ptr(scratch), nb,
                                          _lib.stream_ptr()), "colvo_conv_wgrad_slabs")


def wgrad_reduce_group(sets) -> None:
    """Second half: sets = [(scratch, dw, db, nsplit, Cout, Ctot)] of up to _lib.WGRAD_GROUP_MAX conv_wgrad_slabs calls; ONE launch adds
    every set's slabs to its dw / db in split order."""
    n = len(sets)
    if not 1 <= n <= _lib.WGRAD_GROUP_MAX:
        raise ValueError(f"wgrad_reduce_group: 1..{_lib.WGRAD_GROUP_MAX} sets")
    arr = (_lib.WgradSlabs * n)()
    keep = []
    for i, (scratch, dw, db, nsplit, cout, ctot) in enumerate(sets):
        _need_cuda(scratch, dw, db)
        arr[i].scratch, arr[i].dw, arr[i].db = scratch.data_ptr(), dw.data_ptr(), _lib.ptr(db)
        arr[i].nsplit, arr[i].Cout, arr[i].Ctot = int(nsplit), int(cout), int(ctot)
        keep += [scratch, dw, db]
    rec = program.recording()
    if rec is not None:
        return rec.add(_lib.CMD_WGRAD_REDUCE_GROUP, None, (), (n,), raw=[(0, C.addressof(arr), (arr, keep))])
    _lib.check(_lib.load().colvo_wgrad_reduce_group(arr, n, _lib.stream_ptr()), "colvo_wgrad_reduce_group")


def pack_weights(w_master: torch.Tensor, dtype: torch.dtype, w_fwd: Optional[torch.Tensor],
                 w_bwd: Optional[torch.Tensor]) -> None:
    """w_master fp32 [Cout, kk, Cin] -> w_fwd [Cout, kk, Cin], w_bwd [Cin, kk(flipped), Cout] in `dtype`."""
    _need_cuda(w_master, w_fwd, w_bwd)
    Cout, kk, Cin = w_master.shape
    lib = _lib.load()
    _lib.check(lib.colvo_pack_weights(dt_code(dtype), _lib.ptr(w_master), Cout, kk, Cin, _lib.ptr(w_fwd),
                                      _lib.ptr(w_bwd), _lib.stream_ptr()), "colvo_pack_weights")


def pack_weights_multi(master: torch.Tensor, table: torch.Tensor, nlayers: int, nblocks: int, dtype: torch.dtype,
                       fwd: Optional[torch.Tensor], bwd: torch.Tensor) -> None:
    """All 3x3 layers of a network in one launch (table layout: include/colvo.h colvo_pack_weights_multi)."""
    _need_cuda(master, table, fwd, bwd)
    lib = _lib.load()
    _lib.check(lib.colvo_pack_weights_multi(dt_code(dtype), _lib.ptr(master), _lib.ptr(table), nlayers, nblocks,
                                            _lib.ptr(fwd), _lib.ptr(bwd), _lib.stream_ptr()), "colvo_pack_weights_multi")


def pack_nchw(srcs: Sequence[torch.Tensor], Cpad: int, dtype: torch.dtype, out: Optional[torch.Tensor] = None
              ) -> torch.Tensor:
    """Concatenate NCHW fp32 tensors along channels into one NHWC feature map with Cpad channels."""
    _need_cuda(*srcs)
    B, _, H, W = srcs[0].shape
    srcs = [s.contiguous() for s in srcs]
    if out is None:
        out = torch.empty(B, H, W, Cpad, device=srcs[0].device, dtype=dtype)
    n = len(srcs)
    rec = program.recording()
    if rec is not None:
        rec.add(_lib.CMD_PACK_NCHW, None, list(srcs) + [None] * (4 - n) + [out],
                [dt_code(dtype)] + [s.shape[1] for s in srcs] + [0] * (4 - n) + [n, B, H, W, Cpad])
        return out
    ptrs = (C.c_void_p * n)(*[s.data_ptr() for s in srcs])
    chans = (C.c_int32 * n)(*[s.shape[1] for s in srcs])
    lib = _lib.load()
    _lib.check(lib.colvo_pack_nchw(dt_code(dtype), ptrs, chans, n, B, H, W, Cpad, _lib.ptr(out), _lib.stream_ptr()),
               "colvo_pack_nchw")
    return out


def unpack_nhwc_grad(dsrc: torch.Tensor, c_begin: int, c_count: int, dst: torch.Tensor, accumulate: bool,
                     by_channel: bool = False) -> None:
    """Channels [c_begin, c_begin + c_count) of an NHWC gradient -> fp32 planes: dst [B,c_count,H,W], or with by_channel
    [c_count,B,1,H,W] (every channel a contiguous [B,1,H,W] tensor of its own: one launch for several consumers)."""
    _need_cuda(dsrc, dst)
    B, H, W, Cpad = dsrc.shape
    flags = int(bool(accumulate)) | (2 if by_channel else 0)
    rec = program.recording()
    if rec is not None:
        return rec.add(_lib.CMD_UNPACK_NHWC_GRAD, None, (dsrc, dst),
                       (dt_code(dsrc.dtype), B, H, W, Cpad, c_begin, c_count, flags))
    lib = _lib.load()
    _lib.check(lib.colvo_unpack_nhwc_grad(dt_code(dsrc.dtype), _lib.ptr(dsrc), B, H, W, Cpad, c_begin, c_count,
                                          _lib.ptr(dst), flags, _lib.stream_ptr()), "colvo_unpack_nhwc_grad")


def relu_bwd_inplace(y: torch.Tensor, dy: torch.Tensor) -> None:
    _need_cuda(y, dy)
    lib = _lib.load()
    _lib.check(lib.colvo_relu_bwd_inplace(dt_code(y.dtype), _lib.ptr(y), _lib.ptr(dy), y.numel(), _lib.stream_ptr()),
               "colvo_relu_bwd_inplace")


def depth_head_fwd(x: torch.Tensor, w: torch.Tensor, bias: torch.Tensor, depth: torch.Tensor) -> None:
    _need_cuda(x, w, bias, depth)
    B, H, W, Cc = x.shape
    rec = program.recording()
    if rec is not None:
        return rec.add(_lib.CMD_DEPTH_HEAD_FWD, None, (x, w, bias, depth), (dt_code(x.dtype), B, H, W, Cc),
                       (MIN_DEPTH, MAX_DEPTH))
    lib = _lib.load()
    _lib.check(lib.colvo_depth_head_fwd(dt_code(x.dtype), _lib.ptr(x), _lib.ptr(w), _lib.ptr(bias), B, H, W, Cc,
                                        MIN_DEPTH, MAX_DEPTH, _lib.ptr(depth), _lib.stream_ptr()), "colvo_depth_head_fwd")


def depth_head_bwd(x, w, depth, d_depth, scratch, dx, dw, db) -> None:
    _need_cuda(x, w, depth, d_depth, scratch, dx, dw, db)
    B, H, W, Cc = x.shape
    rec = program.recording()
    if rec is not None:
        return rec.add(_lib.CMD_DEPTH_HEAD_BWD, None, (x, w, depth, d_depth, scratch, dx, dw, db),
                       (dt_code(x.dtype), B, H, W, Cc), (MIN_DEPTH, MAX_DEPTH))
    lib = _lib.load()
    _lib.check(lib.colvo_depth_head_bwd(dt_code(x.dtype), _lib.ptr(x), _lib.ptr(w), _lib.ptr(depth), _lib.ptr(d_depth),
                                        B, H, W, Cc, MIN_DEPTH, MAX_DEPTH, _lib.ptr(scratch), _lib.ptr(dx), _lib.ptr(dw),
                                        _lib.ptr(db), _lib.stream_ptr()), "colvo_depth_head_bwd")


def depth_head_bwd_parts(x, w, depth, g_first, g_second, g_raw, scale_a, scale_b, scratch, dx, g_raw_second=None) -> None:
    """depth_head_bwd with the incoming gradient in parts (DCDP step, depth of 2*Bh images): with s = scale_a[0]*scale_b[0],
    first half g_first + s*g_raw, second half g_second + s*g_raw_second; any part may be None.  Weight gradient:
    depth_head_wgrad."""
    _need_cuda(x, w, depth, g_first, g_second, g_raw, scale_a, scale_b, scratch, dx, g_raw_second)
    B, H, W, Cc = x.shape
    rec = program.recording()
    if rec is not None:
        return rec.add(_lib.CMD_DEPTH_HEAD_BWD_PARTS, None,
                       (x, w, depth, g_first, g_second, g_raw, scale_a, scale_b, scratch, dx, g_raw_second),
                       (dt_code(x.dtype), B, H, W, Cc), (MIN_DEPTH, MAX_DEPTH))
    lib = _lib.load()
    _lib.check(lib.colvo_depth_head_bwd_parts(dt_code(x.dtype), _lib.ptr(x), _lib.ptr(w), _lib.ptr(depth), _lib.ptr(g_first),
                                              _lib.ptr(g_second), _lib.ptr(g_raw), _lib.ptr(g_raw_second), _lib.ptr(scale_a),
                                              _lib.ptr(scale_b),
                                              B, H, W, Cc, MIN_DEPTH, MAX_DEPTH, _lib.ptr(scratch), _lib.ptr(dx), 0, 0,
                                              _lib.stream_ptr()), "colvo_depth_head_bwd_parts")


def zero_multi(tensors) -> None:
    """Zero up to four contiguous device tensors (16-byte aligned, sizes multiples of 16 bytes: the gradient arenas) in ONE
    launch on the current stream."""
    tensors = [t for t in tensors if t is not None and t.numel()]
    _need_cuda(*tensors)
    if not tensors:
        return
    if len(tensors) > _lib.MAX_ARENAS or any(not t.is_contiguous() or (t.numel() * t.element_size()) % 16 or t.data_ptr() % 16
                                             for t in tensors):
        for t in tensors:
            zero_(t)
        return
    n = len(tensors)
    ptrs = (C.c_void_p * n)(*[t.data_ptr() for t in tensors])
    sizes = (C.c_size_t * n)(*[t.numel() * t.element_size() for t in tensors])
    _lib.check(_lib.load().colvo_zero_multi(ptrs, sizes, n, _lib.stream_ptr()), "colvo_zero_multi")


def adam_step_multi(arenas, t: int, *, lr, beta1, beta2, eps, grad_scale=1.0) -> None:
    """Adam over several (param, grad, exp_avg, exp_avg_sq) arena quadruples with one host-side step number, one launch."""
    n = len(arenas)
    arr = (_lib.AdamArena * n)()
    for i, (p, g, m, v) in enumerate(arenas):
        _need_cuda(p, g, m, v)
        arr[i].param, arr[i].grad, arr[i].exp_avg, arr[i].exp_avg_sq, arr[i].n = p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), p.numel()
    _lib.check(_lib.load().colvo_adam_step_multi(arr, n, lr, beta1, beta2, eps, grad_scale, int(t), _lib.stream_ptr()),
               "colvo_adam_step_multi")


def zero_(t: torch.Tensor) -> None:
    """Zero a contiguous device tensor with one hipMemsetAsync on the current stream."""
    _need_cuda(t)
    if not t.is_contiguous():
        raise ValueError("zero_: tensor must be contiguous")
    lib = _lib.load()
    _lib.check(lib.colvo_zero(_lib.ptr(t), t.numel() * t.element_size(), _lib.stream_ptr()), "colvo_zero")


_HEAD_WGRAD_TABLE = _lib.dev_env("COLVO_HEAD_WGRAD_ATOMICS") is None      # developer A/B switch (COLVO_DEV=1)


def depth_head_wgrad(x, dpre, dw, db, deterministic: bool = False) -> None:
    """Weight / bias gradient of the depth head from the d(pre) plane depth_head_bwd(dw=None, db=None) left in scratch.
    The 16-channel head always takes the table form (colvo_depth_head_wgrad_det: per-workgroup partial sums + a fixed-order second
    launch): it is reproducible AND faster than 1536 workgroups' atomics on the same 145 addresses (26 + 5 us against 38)."""
    _need_cuda(x, dpre, dw, db)
    B, H, W, Cc = x.shape
    lib = _lib.load()
    scr, nb = None, 0
    if deterministic or _HEAD_WGRAD_TABLE:
        nb = lib.colvo_depth_head_wgrad_scratch_bytes(B, H, W, Cc)
        if nb == 0 and deterministic:
            raise RuntimeError("deterministic depth-head weight gradient: only the 16-channel head is supported")
    if nb:
        scr = torch.empty((nb + 3) // 4, device=x.device, dtype=torch.float32)
    rec = program.recording()
    if rec is not None:
        return rec.add(_lib.CMD_DEPTH_HEAD_WGRAD, None, (x, dpre, dw, db, scr), (dt_code(x.dtype), B, H, W, Cc, nb))
    if scr is not None:
        _lib.check(lib.colvo_depth_head_wgrad_det(dt_code(x.dtype), _lib.ptr(x), _lib.ptr(dpre), B, H, W, Cc, _lib.ptr(dw),
                                                  _lib.ptr(db), _lib.ptr(scr), nb, _lib.stream_ptr()), "colvo_depth_head_wgrad_det")
        return
    _lib.check(lib.colvo_depth_head_wgrad(dt_code(x.dtype), _lib.ptr(x), _lib.ptr(dpre), B, H, W, Cc, _lib.ptr(dw),
                                          _lib.ptr(db), _lib.stream_ptr()), "colvo_depth_head_wgrad")


def pose_head_fwd(x: torch.Tensor, w: torch.Tensor, bias: torch.Tensor, out: torch.Tensor) -> None:
    """out: 8*B floats, planar [pose Bx6 | lcc_a B | lcc_b B]."""
    _need_cuda(x, w, bias, out)
    B, H, W, Cc = x.shape
    rec = program.recording()
    if rec is not None:
        return rec.add(_lib.CMD_POSE_HEAD_FWD, None, (x, w, bias, out), (dt_code(x.dtype), B, H * W, Cc),
                       (POSE_SCALE, LCC_SCALE))
    lib = _lib.load()
    _lib.check(lib.colvo_pose_head_fwd(dt_code(x.dtype), _lib.ptr(x), _lib.ptr(w), _lib.ptr(bias), B, H * W, Cc,
                                       POSE_SCALE, LCC_SCALE, _lib.ptr(out), _lib.stream_ptr()), "colvo_pose_head_fwd")


def pose_head_bwd(x, w, d_pose, d_a, d_b, dx, dw, db, scale_a=None, scale_b=None, deterministic: bool = False) -> None:
    """d_pose [B,6] / d_a [B,1] / d_b [B,1] contiguous or None (= zero); scale_a, scale_b: device scalars multiplied into
    all three (None = 1).  deterministic: the atomics-free form (colvo_pose_head_bwd_det)."""
    _need_cuda(x, w, d_pose, d_a, d_b, dx, dw, db, scale_a, scale_b)
    B, H, W, Cc = x.shape
    rec = program.recording()
    if rec is not None:
        return rec.add(_lib.CMD_POSE_HEAD_BWD, None, (x, w, d_pose, d_a, d_b, dx, dw, db, scale_a, scale_b),
                       (dt_code(x.dtype), B, H * W, Cc, int(deterministic)), (POSE_SCALE, LCC_SCALE))
    lib = _lib.load()
    fn = lib.colvo_pose_head_bwd_det if deterministic else lib.colvo_pose_head_bwd
    _lib.check(fn(dt_code(x.dtype), _lib.ptr(x), _lib.ptr(w), _lib.ptr(d_pose), _lib.ptr(d_a),
                  _lib.ptr(d_b), _lib.ptr(scale_a), _lib.ptr(scale_b), B, H * W, Cc, POSE_SCALE,
                  LCC_SCALE, _lib.ptr(dx), _lib.ptr(dw), _lib.ptr(db), _lib.stream_ptr()),
               "colvo_pose_head_bwd")


def adam_step_t(param, grad, exp_avg, exp_avg_sq, t: int, *, lr, beta1, beta2, eps, grad_scale=1.0) -> None:
    """Adam with the 1-based step number from the host (colvo_adam_step_t): one launch, no device counter."""
    _need_cuda(param, grad, exp_avg, exp_avg_sq)
    lib = _lib.load()
    _lib.check(lib.colvo_adam_step_t(_lib.ptr(param), _lib.ptr(grad), _lib.ptr(exp_avg), _lib.ptr(exp_avg_sq), param.numel(),
                                     lr, beta1, beta2, eps, grad_scale, int(t), _lib.stream_ptr()), "colvo_adam_step_t")


def adam_step(param, grad, exp_avg, exp_avg_sq, step_count, *, lr, beta1, beta2, eps, grad_scale=1.0) -> None:
    _need_cuda(param, grad, exp_avg, exp_avg_sq, step_count)
    lib = _lib.load()
    _lib.check(lib.colvo_adam_step(_lib.ptr(param), _lib.ptr(grad), _lib.ptr(exp_avg), _lib.ptr(exp_avg_sq),
                                   param.numel(), lr, beta1, beta2, eps, grad_scale, _lib.ptr(step_count),
                                   _lib.stream_ptr()), "colvo_adam_step")
