"""Seeded synthetic frame pairs -- the only data this repo benchmarks on.

The upstream dataset is an external download (/root/reference/README.md:13) and is
not available, so frames are analytic: the target frame is a smooth random field
(8 random 2-D sinusoids per channel + a little noise); the reference frame is the
same field evaluated at flow-shifted coordinates (flow from a smooth ground-truth
depth in [0.5, 5] and a small random pose), then brightness-perturbed with a
per-frame affine map so that the LCC path has something to calibrate.

Plain torch on whatever device is asked for; it is input plumbing, not part of the
measured path.
"""
from __future__ import annotations

import math
from typing import Dict

import torch


def intrinsics(B: int, H: int, W: int, device="cpu", dtype=torch.float32) -> torch.Tensor:
    K = torch.zeros(B, 3, 3, device=device, dtype=dtype)
    K[:, 0, 0] = 0.8 * W
    K[:, 1, 1] = 0.8 * W
    K[:, 0, 2] = (W - 1) / 2.0
    K[:, 1, 2] = (H - 1) / 2.0
    K[:, 2, 2] = 1.0
    return K


def _field(params, x, y):
    """sum_k amp_k * sin(fx_k x + fy_k y + ph_k) per channel, mapped into [0.15, 0.85]."""
    amp, fx, fy, ph = params  # each [B,3,8]
    a = amp[..., None, None]
    arg = fx[..., None, None] * x[:, None, None] + fy[..., None, None] * y[:, None, None] + ph[..., None, None]
    s = (a * torch.sin(arg)).sum(dim=2) / amp.sum(dim=2)[..., None, None]
    return 0.5 + 0.35 * s


def make_batch(B: int, H: int, W: int, seed: int = 1234, device="cpu", dtype=torch.float32
               ) -> Dict[str, torch.Tensor]:
    """-> dict(tgt, ref [B,3,H,W] in [0,1]; K [B,3,3]; gt_depth [B,1,H,W]; gt_pose [B,6]; gt_a, gt_b [B,1])."""
    g = torch.Generator(device="cpu").manual_seed(seed)

    def rnd(*shape):
        return torch.rand(*shape, generator=g, dtype=torch.float64)

    def nrm(*shape):
        return torch.randn(*shape, generator=g, dtype=torch.float64)

    amp = 0.3 + rnd(B, 3, 8)
    ang = 2 * math.pi * rnd(B, 3, 8)
    freq = 2 * math.pi * (1.0 + 7.0 * rnd(B, 3, 8)) / W      # 1..8 periods across the width
    fx, fy = freq * torch.cos(ang), freq * torch.sin(ang)
    ph = 2 * math.pi * rnd(B, 3, 8)

    v, u = torch.meshgrid(torch.arange(H, dtype=torch.float64), torch.arange(W, dtype=torch.float64), indexing="ij")
    u = u.expand(B, H, W)
    v = v.expand(B, H, W)

    # smooth ground-truth depth in [0.5, 5.0]
    dph = 2 * math.pi * rnd(B, 4)
    dfr = 2 * math.pi * (0.5 + rnd(B, 4)) / W
    s = (torch.sin(dfr[:, 0, None, None] * u + dph[:, 0, None, None])
         * torch.cos(dfr[:, 1, None, None] * v + dph[:, 1, None, None]))
    depth = 2.75 + 2.25 * s

    pose = torch.cat([0.02 * nrm(B, 3), 0.01 * nrm(B, 3)], dim=1)
    a = 0.9 + 0.2 * rnd(B, 1)
    b = -0.05 + 0.1 * rnd(B, 1)

    K = intrinsics(B, H, W, dtype=torch.float64)
    f, cx, cy = K[:, 0, 0, None, None], K[:, 0, 2, None, None], K[:, 1, 2, None, None]
    X, Y, Z = (u - cx) / f * depth, (v - cy) / f * depth, depth
    rx, ry, rz = pose[:, 3, None, None], pose[:, 4, None, None], pose[:, 5, None, None]
    tx, ty, tz = pose[:, 0, None, None], pose[:, 1, None, None], pose[:, 2, None, None]
    # small-angle rotation is enough for data generation
    Px = X - rz * Y + ry * Z + tx
    Py = rz * X + Y - rx * Z + ty
    Pz = -ry * X + rx * Y + Z + tz
    du = f * Px / Pz + cx - u
    dv = f * Py / Pz + cy - v

    params = (amp, fx, fy, ph)
    tgt = _field(params, u, v)
    # I_ref(p + flow(p)) = I_tgt(p)  =>  I_ref(q) ~= I_tgt(q - flow(q)); then un-calibrate brightness
    ref = (_field(params, u - du, v - dv) - b[:, :, None, None]) / a[:, :, None, None]
    noise_t = 0.02 * (rnd(B, 3, H, W) - 0.5)
    noise_r = 0.02 * (rnd(B, 3, H, W) - 0.5)
    tgt = (tgt + noise_t).clamp(0, 1)
    ref = (ref + noise_r).clamp(0, 1)

    out = dict(tgt=tgt, ref=ref, K=K, gt_depth=depth.unsqueeze(1), gt_pose=pose, gt_a=a, gt_b=b)
    return {k: t.to(device=device, dtype=dtype).contiguous() for k, t in out.items()}
