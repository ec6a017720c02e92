"""Loss-side operators of the hot path with the signatures of the frozen spec
(oracle/colvo_spec.py: photometric_loss, inverse_warp), backed by the fused HIP kernels.

Concepts: /root/reference/README.md:1 (photometric consistency), :7 (geometric projection between
consecutive frames), :5/:7 (LCC).  The reference ships no code, so the signatures are
the spec's.
"""
from __future__ import annotations

import torch

from . import _lib

import os

SSIM_WEIGHT = 0.85
# Training calls (any of depth / pose / lcc_a / lcc_b requires grad) compute the loss AND its unnormalised gradients in
# one pass (colvo_warp_loss_fused); backward then only applies dL/dloss / max(3 n_valid, 1).  COLVO_LOSS_UNFUSED=1 keeps
# the two-pass form (forward kernel, then a backward kernel that re-evaluates the warp).
FUSE_TRAINING_PASS = _lib.dev_env("COLVO_LOSS_UNFUSED") is None        # developer switch (COLVO_DEV=1)

# Optional kernel timing for bench.py: when enabled, every fused-op call is bracketed by HIP events recorded on
# the stream the kernels are launched on (PyTorch's current stream), immediately around the C-ABI call.
_timing = None


def enable_timing(on: bool = True) -> None:
    global _timing
    _timing = {"fwd": [], "bwd": []} if on else None


def timing_events():
    return _timing


def _timed(kind, call):
    if _timing is None:
        return call()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    r = call()
    e1.record()
    _timing[kind].append((e0, e1))
    return r


# Data parallel (ddp.GradBuckets(exact_batch_loss=True)): called as reducer(state, can_defer) with the loss state {loss, 1/max(3n,1),
# n_valid, masked sum} right after the forward kernel; it adds the last two up over the ranks and rescales (include/colvo.h
# colvo_warp_loss_rescale), so that loss AND gradients are those of the spec's ONE masked mean over the whole batch (oracle/SPEC.md
# section 5) instead of the mean of per-rank means.  Returns None when the state holds the global scale on return (the exchange was
# waited for), or -- only if can_defer: every gradient of this call leaves through the hand-over mailboxes -- a 1-element device
# tensor to hand the gradients' consumers IN PLACE of state[1]: the reducer has then only STARTED the exchange and applies the
# global scale itself, later (ddp.GradBuckets(defer_loss_normalisation=True): a scale of one now, the real one in the optimizer).
# None: single process.
_batch_reducer = None


def set_batch_reducer(fn) -> None:
    global _batch_reducer
    _batch_reducer = fn


class GradHandover:
    """Mailbox between photometric_loss and the producer of one of its arguments (DepthNet.forward_pair_split for `depth`,
    PoseNet.forward for `pose` / `lcc_a` / `lcc_b`).

    The fused loss kernel yields its gradients UNNORMALISED: 1 / max(3 n_valid, 1) is only known once every strip of every
    image has reported, and dL/dloss only at backward time.  Applying them costs a pass over d_depth (84 MB at BASELINE
    configs[2]) and a launch for the pose / LCC gradients -- unless the consumer of those gradients multiplies while it
    reads them anyway.  A tensor that carries a GradHandover (attribute `_colvo_handover`) promises exactly that:
    photometric_loss's backward then returns the unnormalised gradient and posts the two device scalars here; the backward
    node that owns the mailbox picks them up (colvo_depth_head_bwd_parts / colvo_pose_head_bwd).  Such a tensor must reach
    photometric_loss directly and be used for nothing else (take() refuses a gradient that was summed with others)."""

    def __init__(self):
        self.scale_a = self.scale_b = None
        self.raws = None
        self.versions = None

    def post(self, raws, scale_a: torch.Tensor, scale_b: torch.Tensor) -> None:
        # The mailbox keeps the raw tensors THEMSELVES, not their addresses: while it holds a reference autograd cannot
        # accumulate a second gradient into one of them in place (its input buffer does `old.add_(new)` only when it is the
        # sole owner, which it would be -- the loss node's saved variables are released by then -- and the address would not
        # change), so a gradient that was summed with others arrives as a NEW tensor and take() sees it.  The version counters
        # catch any other in-place update.
        self.raws, self.versions = tuple(raws), tuple(r._version for r in raws)
        self.scale_a, self.scale_b = scale_a, scale_b

    def take(self, grads):
        """-> (scale_a, scale_b) for the incoming gradients `grads` (tuple, in post order); (None, None) when nothing was
        posted, i.e. the gradients are ordinary ones."""
        if self.raws is None:
            return None, None
        ok = len(grads) == len(self.raws) and all(
            g is not None and g.data_ptr() == r.data_ptr() and g.shape == r.shape and g._version == v and r._version == v
            for g, r, v in zip(grads, self.raws, self.versions))
        raws, self.raws, self.versions = self.raws, None, None
        if not ok:
            self.scale_a = self.scale_b = None
            raise RuntimeError("a tensor reserved for photometric_loss (DepthNet.forward_pair_split's third output, or "
                               "PoseNet's outputs) was also used elsewhere: its gradient arrived summed with others, which "
                               "the deferred normalisation cannot undo; route other uses through an ordinary tensor "
                               "(e.g. `pose * 1`) so that the loss takes its general path")
        a, b = self.scale_a, self.scale_b
        self.scale_a = self.scale_b = None
        return a, b


def _chk(t: torch.Tensor, name: str, shape) -> torch.Tensor:
    if not t.is_cuda:
        raise RuntimeError(f"{name}: coivo_amd ops run on the GPU only (got a {t.device} tensor); "
                           "there is no CPU fallback")
    if t.dtype != torch.float32:
        raise TypeError(f"{name}: expected float32, got {t.dtype}")
    if tuple(t.shape) != tuple(shape):
        raise ValueError(f"{name}: expected shape {tuple(shape)}, got {tuple(t.shape)}")
    return t.contiguous()


class _WarpLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, tgt, ref, depth, pose, K, lcc_a, lcc_b, ssim_weight, handover, pose_handover):
        lib = _lib.load()
        B, C, H, W = tgt.shape
        if C != 3:
            raise ValueError("photometric_loss: images must have 3 channels")
        tgt = _chk(tgt, "tgt", (B, 3, H, W))
        ref = _chk(ref, "ref", (B, 3, H, W))
        depth = _chk(depth, "depth", (B, 1, H, W))
        pose = _chk(pose, "pose", (B, 6))
        K = _chk(K, "K", (B, 3, 3))
        lcc_a = _chk(lcc_a, "lcc_a", (B, 1))
        lcc_b = _chk(lcc_b, "lcc_b", (B, 1))
        nws = lib.colvo_warp_loss_workspace_floats(B, H, W)
        ws = torch.empty(nws, device=tgt.device, dtype=torch.float32)
        state = torch.empty(4, device=tgt.device, dtype=torch.float32)
        ctx.ssim_weight = float(ssim_weight)
        ctx.fused = FUSE_TRAINING_PASS and any(ctx.needs_input_grad[2:7])
        ctx.handover = handover if (ctx.fused and ctx.needs_input_grad[2]) else None
        # the pose / LCC hand-over rides on the depth one (otherwise the general scaling kernel runs anyway)
        ctx.pose_handover = pose_handover if (ctx.handover is not None and all(ctx.needs_input_grad[i] for i in (3, 5, 6))) else None
        if ctx.fused:
            d_raw = torch.empty_like(depth)
            gpart = torch.empty(B * 14, device=tgt.device, dtype=torch.float32)
            gunit = torch.empty(B * 8, device=tgt.device, dtype=torch.float32) if ctx.pose_handover is not None else None
            _lib.check(_timed("fwd", lambda: lib.colvo_warp_loss_fused(
                _lib.ptr(tgt), _lib.ptr(ref), _lib.ptr(depth), _lib.ptr(pose), _lib.ptr(K), _lib.ptr(lcc_a), _lib.ptr(lcc_b),
                B, H, W, float(ssim_weight), _lib.ptr(ws), _lib.ptr(state), _lib.ptr(d_raw), _lib.ptr(gpart), _lib.ptr(gunit),
                _lib.stream_ptr())), "colvo_warp_loss_fused")
            ctx.scale_b = None
            if _batch_reducer is not None:
                ctx.scale_b = _batch_reducer(state, ctx.pose_handover is not None)
            ctx.gunit = gunit
            # (the state is kept as a plain attribute: a deferred exchange over a process group that completes on a thread of its own
            #  -- gloo -- bumps its version counter AFTER this point, which save_for_backward would report as an in-place modification)
            ctx.state = state
            ctx.save_for_backward(pose, d_raw, gpart)
            ctx.shape = (B, H, W)
            return state[0]
        _lib.check(_timed("fwd", lambda: lib.colvo_warp_loss_fwd(
            _lib.ptr(tgt), _lib.ptr(ref), _lib.ptr(depth), _lib.ptr(pose), _lib.ptr(K), _lib.ptr(lcc_a), _lib.ptr(lcc_b),
            B, H, W, float(ssim_weight), _lib.ptr(ws), _lib.ptr(state), _lib.stream_ptr())), "colvo_warp_loss_fwd")
        if _batch_reducer is not None and any(ctx.needs_input_grad[2:7]):
            _batch_reducer(state, False)
        ctx.save_for_backward(tgt, ref, depth, pose, K, lcc_a, lcc_b, state)
        return state[0]

    @staticmethod
    def backward(ctx, grad_loss):
        lib = _lib.load()
        g = grad_loss.to(torch.float32).contiguous().reshape(1)
        if ctx.fused:
            pose, d_raw, gpart = ctx.saved_tensors
            state = ctx.state
            B, H, W = ctx.shape
            if ctx.pose_handover is not None:
                # everything is handed over unnormalised: this backward launches nothing
                gu = ctx.gunit
                d_pose, d_a, d_b = gu[:6 * B].view(B, 6), gu[6 * B:7 * B].view(B, 1), gu[7 * B:].view(B, 1)
                sb = state[1:2] if ctx.scale_b is None else ctx.scale_b     # (deferred data-parallel normalisation: a scale of one)
                ctx.handover.post((d_raw,), g, sb)
                ctx.pose_handover.post((d_pose, d_a, d_b), g, sb)
                if _timing is not None:
                    _timed("bwd", lambda: 0)
                return None, None, d_raw, d_pose, None, d_a, d_b, None, None, None
            d_pose = torch.empty_like(pose)
            d_a = torch.empty(B, 1, device=pose.device, dtype=torch.float32)
            d_b = torch.empty(B, 1, device=pose.device, dtype=torch.float32)
            if ctx.handover is not None:
                # deferred normalisation: the consumer of d_depth (DepthNet's head backward) applies dL/dloss / max(3n, 1)
                _lib.check(_timed("bwd", lambda: lib.colvo_warp_loss_fused_bwd_params(
                    _lib.ptr(state), _lib.ptr(g), _lib.ptr(gpart), _lib.ptr(pose), B, _lib.ptr(d_pose), _lib.ptr(d_a),
                    _lib.ptr(d_b), _lib.stream_ptr())), "colvo_warp_loss_fused_bwd_params")
                ctx.handover.post((d_raw,), g, state[1:2])
                return None, None, d_raw, d_pose, None, d_a, d_b, None, None, None
            d_depth = torch.empty_like(d_raw)
            _lib.check(_timed("bwd", lambda: lib.colvo_warp_loss_fused_bwd(
                _lib.ptr(state), _lib.ptr(g), _lib.ptr(d_raw), _lib.ptr(gpart), _lib.ptr(pose), B, H, W, _lib.ptr(d_depth),
                _lib.ptr(d_pose), _lib.ptr(d_a), _lib.ptr(d_b), _lib.stream_ptr())), "colvo_warp_loss_fused_bwd")
            return None, None, d_depth, d_pose, None, d_a, d_b, None, None, None
        tgt, ref, depth, pose, K, lcc_a, lcc_b, state = ctx.saved_tensors
        B, _, H, W = tgt.shape
        nws = lib.colvo_warp_loss_workspace_floats(B, H, W)
        ws = torch.empty(nws, device=tgt.device, dtype=torch.float32)
        d_depth = torch.empty_like(depth)
        d_pose = torch.empty_like(pose)
        d_a = torch.empty_like(lcc_a)
        d_b = torch.empty_like(lcc_b)
        _lib.check(_timed("bwd", lambda: lib.colvo_warp_loss_bwd(
            _lib.ptr(tgt), _lib.ptr(ref), _lib.ptr(depth), _lib.ptr(pose), _lib.ptr(K), _lib.ptr(lcc_a), _lib.ptr(lcc_b),
            B, H, W, ctx.ssim_weight, _lib.ptr(state), _lib.ptr(g), _lib.ptr(ws), _lib.ptr(d_depth), _lib.ptr(d_pose),
            _lib.ptr(d_a), _lib.ptr(d_b), _lib.stream_ptr())), "colvo_warp_loss_bwd")
        return None, None, d_depth, d_pose, None, d_a, d_b, None, None, None


def photometric_loss(tgt, ref, depth, pose, K, lcc_a, lcc_b, *, ssim_weight: float = SSIM_WEIGHT) -> torch.Tensor:
    """Masked mean of alpha*(1-SSIM)/2 + (1-alpha)*|I_t - (a*warp(I_r)+b)| -> scalar (spec: photometric_loss).

    When `depth` is the tensor DepthNet.forward_pair_split reserves for this loss (it carries a GradHandover), the
    normalisation of its gradient is left to DepthNet's head backward instead of a separate pass over d_depth."""
    hp = getattr(pose, "_colvo_handover", None)
    if hp is not None and not (getattr(lcc_a, "_colvo_handover", None) is hp and getattr(lcc_b, "_colvo_handover", None) is hp):
        hp = None                      # pose, lcc_a and lcc_b must come from ONE PoseNet call
    # `depth` = the target half `d[:B]` of depth_net(cat(tgt, ref)) (nn.DepthNet.forward): the pass made a second output for
    # exactly this call -- the same values, a gradient path of its own with the deferred normalisation -- so that the spec's
    # call sequence, which hands ONE tensor to PoseNet and to the loss, runs like forward_pair_split.  Taken once; a second loss
    # on the same tensor, or a tensor changed in place since, takes the ordinary path.
    twin = getattr(depth, "_colvo_loss_twin", None)
    if twin is not None:
        depth._colvo_loss_twin = None
        if twin[0]._version == twin[1] and torch.is_grad_enabled():
            depth = twin[0]
    return _WarpLoss.apply(tgt, ref, depth, pose, K, lcc_a, lcc_b, ssim_weight, getattr(depth, "_colvo_handover", None), hp)


# ---------------------------------------------------------------------------------------------------------------------- #
# the widened objective (spec: geometric_consistency_loss, smoothness_loss,                      #
# multiscale_photometric_loss, dcdp_full_loss).  Optional -- BASELINE.json's metric is the plain photometric step.        #
# ---------------------------------------------------------------------------------------------------------------------- #
GEO_WEIGHT, SMOOTH_WEIGHT, NUM_SCALES = 0.5, 0.1, 3


class _GeoLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, depth_t, depth_r, pose, K):
        lib = _lib.load()
        B, _, H, W = depth_t.shape
        depth_t = _chk(depth_t, "tgt_depth", (B, 1, H, W))
        depth_r = _chk(depth_r, "ref_depth", (B, 1, H, W))
        pose = _chk(pose, "pose", (B, 6))
        K = _chk(K, "K", (B, 3, 3))
        ws = torch.empty(lib.colvo_geo_loss_workspace_floats(B, H, W), device=depth_t.device, dtype=torch.float32)
        state = torch.empty(4, device=depth_t.device, dtype=torch.float32)
        _lib.check(lib.colvo_geo_loss_fwd(_lib.ptr(depth_t), _lib.ptr(depth_r), _lib.ptr(pose), _lib.ptr(K), B, H, W,
                                          _lib.ptr(ws), _lib.ptr(state), _lib.stream_ptr()), "colvo_geo_loss_fwd")
        ctx.save_for_backward(depth_t, depth_r, pose, K, state)
        return state[0]

    @staticmethod
    def backward(ctx, grad_loss):
        lib = _lib.load()
        depth_t, depth_r, pose, K, state = ctx.saved_tensors
        B, _, H, W = depth_t.shape
        g = grad_loss.to(torch.float32).contiguous().reshape(1)
        ws = torch.empty(lib.colvo_geo_loss_workspace_floats(B, H, W), device=depth_t.device, dtype=torch.float32)
        d_t, d_r, d_pose = torch.empty_like(depth_t), torch.empty_like(depth_r), torch.empty_like(pose)
        _lib.check(lib.colvo_geo_loss_bwd(_lib.ptr(depth_t), _lib.ptr(depth_r), _lib.ptr(pose), _lib.ptr(K), B, H, W,
                                          _lib.ptr(state), _lib.ptr(g), _lib.ptr(ws), _lib.ptr(d_t), _lib.ptr(d_r),
                                          _lib.ptr(d_pose), _lib.stream_ptr()), "colvo_geo_loss_bwd")
        return d_t, d_r, d_pose, None


def geometric_consistency_loss(tgt_depth, ref_depth, pose, K) -> torch.Tensor:
    """Masked mean of |D_proj - D_samp| / (D_proj + D_samp) -> scalar (spec: geometric_consistency_loss; README.md:1, :7).
    Gradient reaches tgt_depth, ref_depth (float atomics: order-dependent in the last bits) and pose."""
    return _GeoLoss.apply(tgt_depth, ref_depth, pose, K)


class _SmoothLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, depth, img):
        lib = _lib.load()
        B, _, H, W = depth.shape
        depth = _chk(depth, "depth", (B, 1, H, W))
        img = _chk(img, "img", (B, 3, H, W))
        ws = torch.empty(2 * B * ((H * W + 255) // 256), device=depth.device, dtype=torch.float32)
        loss = torch.empty(1, device=depth.device, dtype=torch.float32)
        _lib.check(lib.colvo_smooth_loss_fwd(_lib.ptr(depth), _lib.ptr(img), B, H, W, _lib.ptr(ws), _lib.ptr(loss),
                                             _lib.stream_ptr()), "colvo_smooth_loss_fwd")
        ctx.save_for_backward(depth, img)
        return loss[0]

    @staticmethod
    def backward(ctx, grad_loss):
        lib = _lib.load()
        depth, img = ctx.saved_tensors
        B, _, H, W = depth.shape
        g = grad_loss.to(torch.float32).contiguous().reshape(1)
        d_depth = torch.empty_like(depth)
        _lib.check(lib.colvo_smooth_loss_bwd(_lib.ptr(depth), _lib.ptr(img), B, H, W, _lib.ptr(g), _lib.ptr(d_depth),
                                             _lib.stream_ptr()), "colvo_smooth_loss_bwd")
        return d_depth, None


def smoothness_loss(depth, img) -> torch.Tensor:
    """Edge-aware first-order smoothness of 1/depth -> scalar (spec: smoothness_loss)."""
    return _SmoothLoss.apply(depth, img)


class _AvgPool2(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        lib = _lib.load()
        if not x.is_cuda or x.dtype != torch.float32 or x.dim() != 4:
            raise RuntimeError("downsample2: expects a float32 CUDA tensor [B,C,H,W] (no CPU fallback)")
        B, C, H, W = x.shape
        if H % 2 or W % 2:
            raise ValueError("downsample2: H and W must be even")
        x = x.contiguous()
        y = torch.empty(B, C, H // 2, W // 2, device=x.device, dtype=torch.float32)
        _lib.check(lib.colvo_avgpool2_fwd(_lib.ptr(x), B * C, H, W, _lib.ptr(y), _lib.stream_ptr()), "colvo_avgpool2_fwd")
        ctx.shape = (B, C, H, W)
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = _lib.load()
        B, C, H, W = ctx.shape
        dy = dy.contiguous()
        dx = torch.empty(B, C, H, W, device=dy.device, dtype=torch.float32)
        _lib.check(lib.colvo_avgpool2_bwd(_lib.ptr(dy), B * C, H, W, _lib.ptr(dx), _lib.stream_ptr()), "colvo_avgpool2_bwd")
        return dx


def downsample2(x: torch.Tensor) -> torch.Tensor:
    """2x2 average pooling (spec: downsample2)."""
    return _AvgPool2.apply(x)


def scale_intrinsics(K: torch.Tensor) -> torch.Tensor:
    """Intrinsics of the 2x2-pooled image (spec: scale_intrinsics; pixel centres on integers)."""
    K2 = K.clone()
    K2[:, 0, 0] = K[:, 0, 0] * 0.5
    K2[:, 1, 1] = K[:, 1, 1] * 0.5
    K2[:, 0, 2] = (K[:, 0, 2] - 0.5) * 0.5
    K2[:, 1, 2] = (K[:, 1, 2] - 0.5) * 0.5
    return K2


def multiscale_photometric_loss(tgt, ref, depth, pose, K, lcc_a, lcc_b, *, num_scales: int = NUM_SCALES,
                                ssim_weight: float = SSIM_WEIGHT) -> torch.Tensor:
    """Mean over scales of the fused photometric loss on 2x2-pooled frames / depth (spec: multiscale_photometric_loss).
    Every scale is one launch of the same fused kernel (algorithmic bytes 60 * pixels / 4^s)."""
    total = photometric_loss(tgt, ref, depth, pose, K, lcc_a, lcc_b, ssim_weight=ssim_weight)
    for _ in range(1, num_scales):
        with torch.no_grad():
            tgt, ref, K = downsample2(tgt), downsample2(ref), scale_intrinsics(K)
        depth = downsample2(depth)
        total = total + photometric_loss(tgt, ref, depth, pose, K, lcc_a, lcc_b, ssim_weight=ssim_weight)
    return total / num_scales


def dcdp_full_loss_composite(tgt, ref, d_t, d_r, pose, K, lcc_a, lcc_b, *, geo_weight: float = GEO_WEIGHT,
                             smooth_weight: float = SMOOTH_WEIGHT, num_scales: int = NUM_SCALES,
                             ssim_weight: float = SSIM_WEIGHT) -> torch.Tensor:
    """dcdp_full_loss assembled from the single-term ops with torch arithmetic (spec: dcdp_full_loss, line by line).  Same
    value and gradients as dcdp_full_loss; ~60 element-wise torch launches of autograd glue per step make it host-bound
    (bench.py --full-loss: 2.27 ms per step) -- kept as the cross-check of the one-node form below."""
    loss = multiscale_photometric_loss(tgt, ref, d_t, pose, K, lcc_a, lcc_b, num_scales=num_scales, ssim_weight=ssim_weight)
    if geo_weight:
        loss = loss + geo_weight * geometric_consistency_loss(d_t, d_r, pose, K)
    if smooth_weight:
        loss = loss + smooth_weight * smoothness_loss(d_t, tgt)
    return loss


class _FullObjective(torch.autograd.Function):
    """The widened objective as ONE autograd node over TWO native calls (include/colvo.h colvo_full_objective_fwd / _bwd):
    forward = smoothness + pyramid + the one-pass photometric kernel per level (the level-0 launch carries the geometric-
    consistency term on the projection and taps it has anyway) + one finalize; backward = one launch that turns the saved raw
    planes and per-image sums into every gradient.  No torch kernel runs in either direction; the gradient w.r.t. the
    reference depth is a fixed-point scatter (bit-reproducible)."""

    @staticmethod
    def forward(ctx, tgt, ref, d_t, d_r, pose, K, lcc_a, lcc_b, geo_weight, smooth_weight, num_scales, ssim_weight):
        lib = _lib.load()
        B, C, H, W = tgt.shape
        if C != 3:
            raise ValueError("dcdp_full_loss: images must have 3 channels")
        if num_scales < 1 or (H % (1 << (num_scales - 1))) or (W % (1 << (num_scales - 1))):
            raise ValueError("dcdp_full_loss: H and W must be divisible by 2^(num_scales - 1)")
        if geo_weight and d_r is None:
            raise ValueError("dcdp_full_loss: the geometric-consistency term (geo_weight != 0) needs the reference depth")
        tgt, ref = _chk(tgt, "tgt", (B, 3, H, W)), _chk(ref, "ref", (B, 3, H, W))
        d_t = _chk(d_t, "tgt_depth", (B, 1, H, W))
        d_r = _chk(d_r, "ref_depth", (B, 1, H, W)) if d_r is not None else None
        pose, K = _chk(pose, "pose", (B, 6)), _chk(K, "K", (B, 3, 3))
        lcc_a, lcc_b = _chk(lcc_a, "lcc_a", (B, 1)), _chk(lcc_b, "lcc_b", (B, 1))
        n = lib.colvo_full_objective_workspace_floats(B, H, W, num_scales)
        if n == 0:
            raise ValueError(f"dcdp_full_loss: unsupported shape B={B} H={H} W={W} num_scales={num_scales} (at most 4 scales)")
        f32 = dict(device=tgt.device, dtype=torch.float32)
        ws, loss = torch.empty(n, **f32), torch.empty((), **f32)
        _lib.check(_timed("fwd", lambda: lib.colvo_full_objective_fwd(
            _lib.ptr(tgt), _lib.ptr(ref), _lib.ptr(d_t), _lib.ptr(d_r), _lib.ptr(pose), _lib.ptr(K), _lib.ptr(lcc_a),
            _lib.ptr(lcc_b), B, H, W, num_scales, float(ssim_weight), float(geo_weight), float(smooth_weight), _lib.ptr(ws),
            _lib.ptr(loss), _lib.stream_ptr())), "colvo_full_objective_fwd")
        ctx.cfg = (B, H, W, num_scales, float(geo_weight), float(smooth_weight))
        ctx.ws = ws
        ctx.save_for_backward(pose)
        return loss

    @staticmethod
    def backward(ctx, grad_loss):
        lib = _lib.load()
        (pose,) = ctx.saved_tensors
        B, H, W, num_scales, geo_weight, smooth_weight = ctx.cfg
        ws, ctx.ws = ctx.ws, None
        if ws is None:
            raise RuntimeError("dcdp_full_loss: backward through the same node twice (its workspace is released after the first)")
        f32 = dict(device=pose.device, dtype=torch.float32)
        g = grad_loss.to(torch.float32).contiguous()
        g_dt = torch.empty(B, 1, H, W, **f32)
        g_dr = torch.empty(B, 1, H, W, **f32) if geo_weight else None
        dp, da, db = torch.empty(B, 6, **f32), torch.empty(B, 1, **f32), torch.empty(B, 1, **f32)
        _lib.check(_timed("bwd", lambda: lib.colvo_full_objective_bwd(
            _lib.ptr(ws), _lib.ptr(g), _lib.ptr(pose), B, H, W, num_scales, geo_weight, smooth_weight, _lib.ptr(g_dt),
            _lib.ptr(g_dr), _lib.ptr(dp), _lib.ptr(da), _lib.ptr(db), _lib.stream_ptr())), "colvo_full_objective_bwd")
        return None, None, g_dt, g_dr, dp, None, da, db, None, None, None, None


def full_objective_terms(loss: torch.Tensor):
    """Not part of the spec: the individual terms of the dcdp_full_loss value `loss` came from, for logging -- a view of 32
    device floats: [0] total, [1] geometric term, [2] smoothness term, [4 + 4 s] photometric loss of pyramid level s."""
    fn = loss.grad_fn
    if fn is None or not hasattr(fn, "ws") or fn.ws is None:
        raise ValueError("full_objective_terms: needs the (not yet back-propagated) output of dcdp_full_loss")
    B, H, W, S, _, _ = fn.cfg
    import ctypes
    state = ctypes.c_void_p()
    _lib.check(_lib.load().colvo_full_objective_terms(_lib.ptr(fn.ws), B, H, W, S, ctypes.byref(state)), "colvo_full_objective_terms")
    off = (state.value - fn.ws.data_ptr()) // 4
    return fn.ws[off:off + 32]


def dcdp_full_loss(tgt, ref, d_t, d_r, pose, K, lcc_a, lcc_b, *, geo_weight: float = GEO_WEIGHT,
                   smooth_weight: float = SMOOTH_WEIGHT, num_scales: int = NUM_SCALES,
                   ssim_weight: float = SSIM_WEIGHT) -> torch.Tensor:
    """Multi-scale photometric + geo_weight * geometric consistency + smooth_weight * smoothness (spec: dcdp_full_loss),
    as one autograd node over two native calls (dcdp_full_loss_composite is the term-by-term form the tests compare it
    with).  d_r may be None when geo_weight == 0."""
    return _FullObjective.apply(tgt, ref, d_t, d_r, pose, K, lcc_a, lcc_b, float(geo_weight), float(smooth_weight),
                                int(num_scales), float(ssim_weight))


def inverse_warp(ref, depth, pose, K):
    """Un-fused debugging entry (spec: inverse_warp) -> (warped [B,C,H,W], valid [B,1,H,W]); no autograd."""
    lib = _lib.load()
    B, C, H, W = ref.shape
    ref = _chk(ref, "ref", (B, C, H, W))
    depth = _chk(depth, "depth", (B, 1, H, W))
    pose = _chk(pose, "pose", (B, 6))
    K = _chk(K, "K", (B, 3, 3))
    warped = torch.empty_like(ref)
    valid = torch.empty_like(depth)
    _lib.check(lib.colvo_inverse_warp(_lib.ptr(ref), _lib.ptr(depth), _lib.ptr(pose), _lib.ptr(K), B, C, H, W,
                                      _lib.ptr(warped), _lib.ptr(valid), _lib.stream_ptr()), "colvo_inverse_warp")
    return warped, valid
