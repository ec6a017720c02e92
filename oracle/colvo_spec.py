"""ColVO DCDP+LCC training hot path -- frozen CPU specification ("the oracle").

TEST INFRASTRUCTURE ONLY.  Nothing under ``coivo_amd/`` may import this module;
only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` do, and only as the checker / the timed CPU baseline.

PARITY UNPINNED.  The upstream reference (``/root/reference``) ships a README
and three figures and *no source code, tests or golden vectors* (SURVEY.md §0,
§8c).  The only statements this file can follow are

  * ``/root/reference/README.md:1``  -- "Visual Odometry Considering Geometric
    and Photometric Consistency"  (a photometric-consistency training loss);
  * ``/root/reference/README.md:5``  -- DCDP "a deep couple strategy for depth
    and pose estimation", LCC "a light consistent calibration mechanism";
  * ``/root/reference/README.md:7``  -- DCDP uses "multimodal fusion and loss
    function constraints to couple depth and pose estimation modes" to align
    "geometric projections between consecutive frames"; LCC works "by
    recalibrating the luminosity values of adjacent frames".

Everything else (layer widths, activation, depth parameterisation, rotation
convention, SSIM window, mask rule, reduction) is an [ASSUMED] builder's choice
in the convention of the self-supervised depth/ego-motion family ColVO compares
itself with (``imgs/trajectorypredictions.png``).  Those choices are written
down once in ``oracle/SPEC.md`` and frozen here; this module *is* the
"reference PyTorch CPU path" that ``BASELINE.json:north_star`` asks parity with
(1e-4 on depth maps, 1e-5 on the scalar loss, fp32).

Only stock ``torch`` ops are used, every function is dtype-generic so that
``torch.autograd.gradcheck`` can run it in fp64.
"""
from __future__ import annotations

import math
from typing import Optional, Tuple

import torch
import torch.nn as nn
import torch.nn.functional as F

# --------------------------------------------------------------------------- #
# Frozen constants (SPEC.md §2)                                               #
# --------------------------------------------------------------------------- #
MIN_DEPTH = 0.1          # depth = 1 / (1/MAX + (1/MIN - 1/MAX) * sigmoid(x))
MAX_DEPTH = 10.0
POSE_SCALE = 0.01        # pose = POSE_SCALE * mean_hw(pred[:, 0:6])
LCC_SCALE = 0.1          # a = 1 + LCC_SCALE * pred[:, 6],  b = LCC_SCALE * pred[:, 7]
SSIM_C1 = 0.01 ** 2
SSIM_C2 = 0.03 ** 2
SSIM_WEIGHT = 0.85       # alpha in  alpha*(1-SSIM)/2 + (1-alpha)*|I_t - I'|
Z_EPS = 1e-3             # a projected point is valid only if z > Z_EPS

ENC_CH = (32, 64, 128, 256, 512)   # DepthNet encoder widths at H/2 .. H/32
DEC_CH = (16, 32, 64, 128, 256)    # DepthNet decoder widths at H .. H/16
POSE_CH = (16, 32, 64, 128, 256, 256, 256)  # PoseNet stride-2 stack


# --------------------------------------------------------------------------- #
# a3: pose vector -> matrix, projection (README.md:7 "geometric projections   #
#     between consecutive frames")                                            #
# --------------------------------------------------------------------------- #
def pose_vec2mat(pose: torch.Tensor) -> torch.Tensor:
    """[B,6] = (tx,ty,tz, rx,ry,rz)  ->  [B,3,4] = [R | t],  R = Rz(rz) Ry(ry) Rx(rx).

    The transform maps a point in the *target* camera frame to the *reference*
    camera frame:  P_ref = R @ P_tgt + t.
    """
    t = pose[:, 0:3].unsqueeze(-1)
    rx, ry, rz = pose[:, 3], pose[:, 4], pose[:, 5]
    cx, sx = torch.cos(rx), torch.sin(rx)
    cy, sy = torch.cos(ry), torch.sin(ry)
    cz, sz = torch.cos(rz), torch.sin(rz)
    # R = Rz @ Ry @ Rx written out (row-major)
    r00 = cz * cy
    r01 = cz * sy * sx - sz * cx
    r02 = cz * sy * cx + sz * sx
    r10 = sz * cy
    r11 = sz * sy * sx + cz * cx
    r12 = sz * sy * cx - cz * sx
    r20 = -sy
    r21 = cy * sx
    r22 = cy * cx
    R = torch.stack([r00, r01, r02, r10, r11, r12, r20, r21, r22], dim=1).view(-1, 3, 3)
    return torch.cat([R, t], dim=2)


def project(depth: torch.Tensor, pose: torch.Tensor, K: torch.Tensor
            ) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """Sampling coordinates of every target pixel in the reference frame.

    depth [B,1,H,W] (>0), pose [B,6], K [B,3,3] (fx,0,cx / 0,fy,cy / 0,0,1; the
    skew and the last row are ignored).  Returns x, y in *pixel* units (pixel
    centres at integer coordinates, column u / row v) and valid [B,H,W] (bool).
    """
    B, _, H, W = depth.shape
    dt, dev = depth.dtype, depth.device
    fx, fy = K[:, 0, 0].view(B, 1, 1), K[:, 1, 1].view(B, 1, 1)
    cx, cy = K[:, 0, 2].view(B, 1, 1), K[:, 1, 2].view(B, 1, 1)
    u = torch.arange(W, dtype=dt, device=dev).view(1, 1, W)
    v = torch.arange(H, dtype=dt, device=dev).view(1, H, 1)
    d = depth[:, 0]
    # back-projection  c = D * K^-1 [u, v, 1]
    X = (u - cx) / fx * d
    Y = (v - cy) / fy * d
    Z = d
    T = pose_vec2mat(pose)
    R, t = T[:, :, :3], T[:, :, 3]

    def row(i):
        return (R[:, i, 0].view(B, 1, 1) * X + R[:, i, 1].view(B, 1, 1) * Y
                + R[:, i, 2].view(B, 1, 1) * Z + t[:, i].view(B, 1, 1))

    Px, Py, Pz = row(0), row(1), row(2)
    front = Pz > Z_EPS
    Pz_safe = torch.where(front, Pz, torch.ones_like(Pz))
    x = fx * Px / Pz_safe + cx
    y = fy * Py / Pz_safe + cy
    valid = front & (x >= 0) & (x <= W - 1) & (y >= 0) & (y <= H - 1)
    return x, y, valid


# --------------------------------------------------------------------------- #
# a4: bilinear sample / inverse warp (view synthesis)                         #
# --------------------------------------------------------------------------- #
def bilinear_sample(img: torch.Tensor, x: torch.Tensor, y: torch.Tensor,
                    valid: torch.Tensor) -> torch.Tensor:
    """img [B,C,H,W], x/y/valid [B,H',W'] -> [B,C,H',W'];  0 where not valid.

    Explicit 4-tap gather so the border rule does not depend on grid_sample
    flags: taps are taken at floor(x), floor(x)+1 (clamped into the image; for a
    valid point every tap with non-zero weight is inside).  Invalid points
    yield 0 and carry no gradient into x / y.
    """
    B, C, H, W = img.shape
    xs = torch.where(valid, x, torch.zeros_like(x)).clamp(0, W - 1)
    ys = torch.where(valid, y, torch.zeros_like(y)).clamp(0, H - 1)
    x0f, y0f = torch.floor(xs), torch.floor(ys)
    wx, wy = (xs - x0f).unsqueeze(1), (ys - y0f).unsqueeze(1)
    x0, y0 = x0f.long(), y0f.long()
    x1, y1 = (x0 + 1).clamp(max=W - 1), (y0 + 1).clamp(max=H - 1)
    flat = img.reshape(B, C, H * W)

    def tap(yy, xx):
        idx = (yy * W + xx).view(B, 1, -1).expand(B, C, -1)
        return torch.gather(flat, 2, idx).view(B, C, *x.shape[1:])

    top = tap(y0, x0) * (1 - wx) + tap(y0, x1) * wx
    bot = tap(y1, x0) * (1 - wx) + tap(y1, x1) * wx
    out = top * (1 - wy) + bot * wy
    return out * valid.unsqueeze(1).to(img.dtype)


def inverse_warp(ref: torch.Tensor, depth: torch.Tensor, pose: torch.Tensor,
                 K: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """Synthesise the target view from ``ref``.  -> (warped [B,C,H,W], valid [B,1,H,W] float)."""
    x, y, valid = project(depth, pose, K)
    warped = bilinear_sample(ref, x, y, valid)
    return warped, valid.unsqueeze(1).to(ref.dtype)


# --------------------------------------------------------------------------- #
# a5: LCC (README.md:5, :7 "recalibrating the luminosity values of adjacent   #
#     frames");  a6: SSIM + L1 photometric consistency (README.md:1)          #
# --------------------------------------------------------------------------- #
def lcc_recalibrate(warped: torch.Tensor, lcc_a: torch.Tensor, lcc_b: torch.Tensor) -> torch.Tensor:
    """Per-frame affine brightness map  I' = a * I_warp + b;  a, b: [B,1]."""
    B = warped.shape[0]
    return lcc_a.view(B, 1, 1, 1) * warped + lcc_b.view(B, 1, 1, 1)


def ssim_dissimilarity(x: torch.Tensor, y: torch.Tensor) -> torch.Tensor:
    """clamp((1 - SSIM(x, y)) / 2, 0, 1) with a 3x3 box window over a 1-px reflection pad."""
    xp = F.pad(x, (1, 1, 1, 1), mode="reflect")
    yp = F.pad(y, (1, 1, 1, 1), mode="reflect")
    mu_x = F.avg_pool2d(xp, 3, 1)
    mu_y = F.avg_pool2d(yp, 3, 1)
    sig_x = F.avg_pool2d(xp * xp, 3, 1) - mu_x * mu_x
    sig_y = F.avg_pool2d(yp * yp, 3, 1) - mu_y * mu_y
    sig_xy = F.avg_pool2d(xp * yp, 3, 1) - mu_x * mu_y
    n = (2 * mu_x * mu_y + SSIM_C1) * (2 * sig_xy + SSIM_C2)
    d = (mu_x * mu_x + mu_y * mu_y + SSIM_C1) * (sig_x + sig_y + SSIM_C2)
    return ((1 - n / d) / 2).clamp(0, 1)


def photometric_loss_map(tgt, ref, depth, pose, K, lcc_a, lcc_b, *, ssim_weight: float = SSIM_WEIGHT):
    """Per-pixel, per-channel loss map [B,C,H,W] and validity mask [B,1,H,W] (debug entry)."""
    warped, valid = inverse_warp(ref, depth, pose, K)
    recal = lcc_recalibrate(warped, lcc_a, lcc_b)
    l1 = (tgt - recal).abs()
    ss = ssim_dissimilarity(tgt, recal)
    return ssim_weight * ss + (1 - ssim_weight) * l1, valid


def photometric_loss(tgt: torch.Tensor, ref: torch.Tensor, depth: torch.Tensor, pose: torch.Tensor,
                     K: torch.Tensor, lcc_a: torch.Tensor, lcc_b: torch.Tensor, *,
                     ssim_weight: float = SSIM_WEIGHT) -> torch.Tensor:
    """Masked mean of  alpha*(1-SSIM)/2 + (1-alpha)*|I_t - (a*warp(I_r) + b)|  -> scalar.

    tgt/ref [B,3,H,W] in [0,1]; depth [B,1,H,W]; pose [B,6]; K [B,3,3]; lcc_a/lcc_b [B,1].
    loss = sum(map * mask) / max(C * sum(mask), 1).
    """
    m, valid = photometric_loss_map(tgt, ref, depth, pose, K, lcc_a, lcc_b, ssim_weight=ssim_weight)
    C = m.shape[1]
    denom = (valid.sum() * C).clamp(min=1.0)
    return (m * valid).sum() / denom


# --------------------------------------------------------------------------- #
# §8f-1: geometric consistency (README.md:1 "Considering Geometric and         #
#        Photometric Consistency", :7 "loss function constraints ... alignment #
#        of geometric projections between consecutive frames")                 #
# --------------------------------------------------------------------------- #
GEO_WEIGHT = 0.5         # weight of the geometric-consistency term in the widened DCDP loss
SMOOTH_WEIGHT = 0.1      # weight of the edge-aware smoothness term
NUM_SCALES = 3           # scales of the multi-scale photometric term (full, 1/2, 1/4)


def geometric_consistency_loss(tgt_depth: torch.Tensor, ref_depth: torch.Tensor, pose: torch.Tensor,
                               K: torch.Tensor) -> torch.Tensor:
    """Masked mean of |D_proj - D_samp| / (D_proj + D_samp) -> scalar.

    Every target pixel is back-projected with `tgt_depth`, moved by `pose` and projected into the reference frame
    (project()).  D_proj is the z of that point in the reference camera; D_samp is the reference frame's OWN depth
    prediction, bilinearly sampled where the point lands (bilinear_sample(), same border / validity rule as the
    image warp).  Both depth maps and the pose receive gradient: through D_proj (depth, pose), through the sample
    position (depth, pose) and through the sampled taps (ref_depth).  tgt_depth, ref_depth [B,1,H,W]; pose [B,6]."""
    B, _, H, W = tgt_depth.shape
    x, y, valid = project(tgt_depth, pose, K)
    fx, fy = K[:, 0, 0].view(B, 1, 1), K[:, 1, 1].view(B, 1, 1)
    cx, cy = K[:, 0, 2].view(B, 1, 1), K[:, 1, 2].view(B, 1, 1)
    dt, dev = tgt_depth.dtype, tgt_depth.device
    u = torch.arange(W, dtype=dt, device=dev).view(1, 1, W)
    v = torch.arange(H, dtype=dt, device=dev).view(1, H, 1)
    d = tgt_depth[:, 0]
    T = pose_vec2mat(pose)
    R, t = T[:, :, :3], T[:, :, 3]
    X, Y = (u - cx) / fx * d, (v - cy) / fy * d
    d_proj = (R[:, 2, 0].view(B, 1, 1) * X + R[:, 2, 1].view(B, 1, 1) * Y + R[:, 2, 2].view(B, 1, 1) * d
              + t[:, 2].view(B, 1, 1))
    d_samp = bilinear_sample(ref_depth, x, y, valid)[:, 0]
    m = valid.to(dt)
    num = (d_proj - d_samp).abs()
    den = torch.where(valid, d_proj + d_samp, torch.ones_like(d_proj))
    return (num / den * m).sum() / m.sum().clamp(min=1.0)


# --------------------------------------------------------------------------- #
# §8f-2: edge-aware smoothness and the multi-scale photometric term           #
#        ([ASSUMED] convention of the method family, SPEC.md §8)              #
# --------------------------------------------------------------------------- #
def smoothness_loss(depth: torch.Tensor, img: torch.Tensor) -> torch.Tensor:
    """Edge-aware first-order smoothness of the inverse depth -> scalar.

    mean_x |disp[x+1] - disp[x]| * exp(-mean_c |I[x+1] - I[x]|)  +  the same along y,  disp = 1 / depth;
    each mean runs over all neighbour pairs of the batch.  depth [B,1,H,W] (> 0), img [B,3,H,W] (no gradient)."""
    disp = 1.0 / depth
    ddx = (disp[..., :, 1:] - disp[..., :, :-1]).abs()
    ddy = (disp[..., 1:, :] - disp[..., :-1, :]).abs()
    wx = torch.exp(-(img[..., :, 1:] - img[..., :, :-1]).abs().mean(dim=1, keepdim=True))
    wy = torch.exp(-(img[..., 1:, :] - img[..., :-1, :]).abs().mean(dim=1, keepdim=True))
    return (ddx * wx).mean() + (ddy * wy).mean()


def downsample2(x: torch.Tensor) -> torch.Tensor:
    """2x2 average pooling (H, W even)."""
    return F.avg_pool2d(x, 2, 2)


def scale_intrinsics(K: torch.Tensor) -> torch.Tensor:
    """Intrinsics of the 2x2-average-pooled image (pixel centres on integers: u = 2 u' + 1/2)."""
    K2 = K.clone()
    K2[:, 0, 0] = K[:, 0, 0] * 0.5
    K2[:, 1, 1] = K[:, 1, 1] * 0.5
    K2[:, 0, 2] = (K[:, 0, 2] - 0.5) * 0.5
    K2[:, 1, 2] = (K[:, 1, 2] - 0.5) * 0.5
    return K2


def multiscale_photometric_loss(tgt, ref, depth, pose, K, lcc_a, lcc_b, *, num_scales: int = NUM_SCALES,
                                ssim_weight: float = SSIM_WEIGHT) -> torch.Tensor:
    """Mean over `num_scales` scales of photometric_loss on 2x2-average-pooled frames, depth and scaled intrinsics
    (scale 0 = full resolution).  H, W divisible by 2^(num_scales-1)."""
    total = photometric_loss(tgt, ref, depth, pose, K, lcc_a, lcc_b, ssim_weight=ssim_weight)
    for _ in range(1, num_scales):
        tgt, ref, depth, K = downsample2(tgt), downsample2(ref), downsample2(depth), scale_intrinsics(K)
        total = total + photometric_loss(tgt, ref, depth, pose, K, lcc_a, lcc_b, ssim_weight=ssim_weight)
    return total / num_scales


def dcdp_full_loss(tgt, ref, d_t, d_r, pose, K, lcc_a, lcc_b, *, geo_weight: float = GEO_WEIGHT,
                   smooth_weight: float = SMOOTH_WEIGHT, num_scales: int = NUM_SCALES,
                   ssim_weight: float = SSIM_WEIGHT) -> torch.Tensor:
    """The widened DCDP objective: multi-scale photometric + geo_weight * geometric consistency
    + smooth_weight * edge-aware smoothness of the target depth."""
    loss = multiscale_photometric_loss(tgt, ref, d_t, pose, K, lcc_a, lcc_b, num_scales=num_scales, ssim_weight=ssim_weight)
    if geo_weight:
        loss = loss + geo_weight * geometric_consistency_loss(d_t, d_r, pose, K)
    if smooth_weight:
        loss = loss + smooth_weight * smoothness_loss(d_t, tgt)
    return loss


# --------------------------------------------------------------------------- #
# a1: DepthNet (encoder-decoder), a2: PoseNet (DCDP coupling + LCC head)      #
# --------------------------------------------------------------------------- #
def disp_to_depth(sig: torch.Tensor) -> torch.Tensor:
    lo, hi = 1.0 / MAX_DEPTH, 1.0 / MIN_DEPTH
    return 1.0 / (lo + (hi - lo) * sig)


def _conv(cin, cout, k=3, stride=1):
    return nn.Conv2d(cin, cout, k, stride=stride, padding=k // 2, bias=True)


def init_weights(module: nn.Module, seed: int) -> None:
    """Deterministic init shared by the oracle and the HIP path: every conv weight
    ~ N(0, 2/fan_in) drawn in registration order from one CPU generator, biases 0."""
    g = torch.Generator(device="cpu").manual_seed(seed)
    with torch.no_grad():
        for name, p in module.named_parameters():
            if name.endswith("weight"):
                fan_in = p.shape[1] * p.shape[2] * p.shape[3]
                w = torch.randn(p.shape, generator=g, dtype=torch.float32) * math.sqrt(2.0 / fan_in)
                p.copy_(w.to(p.dtype))
            else:
                p.zero_()


class DepthNet(nn.Module):
    """BN-free U-Net.  forward(img [B,3,H,W]) -> depth [B,1,H,W] in (MIN_DEPTH, MAX_DEPTH).

    H and W must be multiples of 32.  Encoder level i (1..5): conv3x3/s2 + ReLU,
    conv3x3/s1 + ReLU.  Decoder level i (5..1): nearest 2x up-sample -> conv3x3 +
    ReLU ("up"), concat with the encoder skip of the same resolution -> conv3x3 +
    ReLU ("iconv").  Head: conv3x3 -> sigmoid -> disp_to_depth.
    """

    def __init__(self):
        super().__init__()
        cin = 3
        for i, c in enumerate(ENC_CH, start=1):
            setattr(self, f"enc{i}a", _conv(cin, c, 3, 2))
            setattr(self, f"enc{i}b", _conv(c, c, 3, 1))
            cin = c
        for i in range(5, 0, -1):
            d = DEC_CH[i - 1]
            setattr(self, f"up{i}", _conv(cin, d, 3, 1))
            skip = ENC_CH[i - 2] if i >= 2 else 0
            setattr(self, f"iconv{i}", _conv(d + skip, d, 3, 1))
            cin = d
        self.head = _conv(cin, 1, 3, 1)

    def forward(self, img: torch.Tensor) -> torch.Tensor:
        skips = []
        x = img
        for i in range(1, 6):
            x = F.relu(getattr(self, f"enc{i}a")(x))
            x = F.relu(getattr(self, f"enc{i}b")(x))
            skips.append(x)
        for i in range(5, 0, -1):
            x = F.interpolate(x, scale_factor=2, mode="nearest")
            x = F.relu(getattr(self, f"up{i}")(x))
            if i >= 2:
                x = torch.cat([x, skips[i - 2]], dim=1)
            x = F.relu(getattr(self, f"iconv{i}")(x))
        return disp_to_depth(torch.sigmoid(self.head(x)))


class PoseNet(nn.Module):
    """DCDP pose network with the LCC head.

    forward(tgt, ref, tgt_depth=None, ref_depth=None) -> (pose [B,6], lcc_a [B,1], lcc_b [B,1]).
    Input = channel concat (tgt, ref, tgt_depth, ref_depth) = 8 channels -- the RGB
    pair fused with both predicted depth maps ("multimodal fusion ... to couple
    depth and pose estimation modes", README.md:7); missing depths are zeros.
    Seven conv3x3/s2 + ReLU, a 1x1 conv to 8 channels, spatial mean.
    """

    def __init__(self):
        super().__init__()
        cin = 8
        for i, c in enumerate(POSE_CH, start=1):
            setattr(self, f"conv{i}", _conv(cin, c, 3, 2))
            cin = c
        self.pred = _conv(cin, 8, 1, 1)

    def forward(self, tgt, ref, tgt_depth: Optional[torch.Tensor] = None,
                ref_depth: Optional[torch.Tensor] = None):
        B, _, H, W = tgt.shape
        z = tgt.new_zeros(B, 1, H, W)
        x = torch.cat([tgt, ref, z if tgt_depth is None else tgt_depth,
                       z if ref_depth is None else ref_depth], dim=1)
        for i in range(1, 8):
            x = F.relu(getattr(self, f"conv{i}")(x))
        o = self.pred(x).mean(dim=(2, 3))
        pose = POSE_SCALE * o[:, 0:6]
        lcc_a = 1.0 + LCC_SCALE * o[:, 6:7]
        lcc_b = LCC_SCALE * o[:, 7:8]
        return pose, lcc_a, lcc_b


# --------------------------------------------------------------------------- #
# the DCDP + LCC training step                                                #
# --------------------------------------------------------------------------- #
def dcdp_forward(depth_net: nn.Module, pose_net: nn.Module, tgt, ref, K, *,
                 ssim_weight: float = SSIM_WEIGHT, full_loss: bool = False):
    """One coupled forward:  depth of both frames -> pose + LCC -> photometric loss
    (full_loss: the widened objective dcdp_full_loss instead -- multi-scale photometric + geometric consistency
    + smoothness; the round-1 hot path and BASELINE.json's metric use the plain photometric loss).

    Returns (loss, tgt_depth, ref_depth, pose, lcc_a, lcc_b).
    """
    B = tgt.shape[0]
    d = depth_net(torch.cat([tgt, ref], dim=0))
    d_t, d_r = d[:B], d[B:]
    pose, a, b = pose_net(tgt, ref, d_t, d_r)
    if full_loss:
        loss = dcdp_full_loss(tgt, ref, d_t, d_r, pose, K, a, b, ssim_weight=ssim_weight)
    else:
        loss = photometric_loss(tgt, ref, d_t, pose, K, a, b, ssim_weight=ssim_weight)
    return loss, d_t, d_r, pose, a, b


def train_step(depth_net, pose_net, optimizer, tgt, ref, K) -> torch.Tensor:
    optimizer.zero_grad(set_to_none=True)
    loss = dcdp_forward(depth_net, pose_net, tgt, ref, K)[0]
    loss.backward()
    optimizer.step()
    return loss.detach()


ADAM_KW = dict(lr=1e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0)


# --------------------------------------------------------------------------- #
# §8f-3: inference path (README.md:9 "complete 3D reconstruction of the        #
#        intestine", :29 "stitching together the dense depth maps of each      #
#        frame using the colonoscopic trajectory")                             #
# --------------------------------------------------------------------------- #
def resize_frames_u8(frames_u8: torch.Tensor, H: int, W: int) -> torch.Tensor:
    """§8f-4 input format: [B,h,w,3] uint8 interleaved RGB -> [B,3,H,W] float32 in [0,1], bilinear, half-pixel centres."""
    x = frames_u8.permute(0, 3, 1, 2).to(torch.float32)
    if tuple(x.shape[-2:]) != (H, W):
        x = F.interpolate(x, size=(H, W), mode="bilinear", align_corners=False, antialias=False)
    return x / 255.0


def resize_intrinsics(K: torch.Tensor, hw_from, hw_to) -> torch.Tensor:
    """Intrinsics of a frame resized by resize_frames_u8: pixel centres on integers (§1), so a point at source
    coordinate x lands at (x + 0.5) * s - 0.5."""
    (h, w), (H, W) = hw_from, hw_to
    sy, sx = H / h, W / w
    K2 = K.clone()
    K2[..., 0, 0] = K[..., 0, 0] * sx
    K2[..., 1, 1] = K[..., 1, 1] * sy
    K2[..., 0, 2] = (K[..., 0, 2] + 0.5) * sx - 0.5
    K2[..., 1, 2] = (K[..., 1, 2] + 0.5) * sy - 0.5
    return K2


def pose_to_matrix4(pose: torch.Tensor) -> torch.Tensor:
    """[N,6] -> [N,4,4] homogeneous transform (P_ref = R P_tgt + t, pose_vec2mat)."""
    T = pose_vec2mat(pose)
    bottom = torch.zeros(T.shape[0], 1, 4, dtype=T.dtype, device=T.device)
    bottom[:, 0, 3] = 1.0
    return torch.cat([T, bottom], dim=1)


def integrate_trajectory(rel_poses: torch.Tensor) -> torch.Tensor:
    """Camera-to-world transforms of frames 0..N from the N relative poses of consecutive pairs.

    rel_poses[k] is PoseNet's output for (target = frame k, reference = frame k+1), i.e. it maps frame-k points into
    frame k+1.  World = camera 0, so  M_0 = I,  M_{k+1} = M_k @ inverse(T_k).  -> [N+1,4,4] (float64 recommended)."""
    T = pose_to_matrix4(rel_poses)
    M = [torch.eye(4, dtype=T.dtype, device=T.device)]
    for k in range(T.shape[0]):
        R, t = T[k, :3, :3], T[k, :3, 3]
        Tinv = torch.eye(4, dtype=T.dtype, device=T.device)
        Tinv[:3, :3] = R.t()
        Tinv[:3, 3] = -(R.t() @ t)
        M.append(M[-1] @ Tinv)
    return torch.stack(M)


def backproject(depth: torch.Tensor, K: torch.Tensor, cam2world: torch.Tensor) -> torch.Tensor:
    """depth [B,1,H,W], K [B,3,3], cam2world [B,4,4] -> world points [B,H*W,3] (row-major pixel order)."""
    B, _, H, W = depth.shape
    dt, dev = depth.dtype, depth.device
    fx, fy = K[:, 0, 0].view(B, 1, 1), K[:, 1, 1].view(B, 1, 1)
    cx, cy = K[:, 0, 2].view(B, 1, 1), K[:, 1, 2].view(B, 1, 1)
    u = torch.arange(W, dtype=dt, device=dev).view(1, 1, W)
    v = torch.arange(H, dtype=dt, device=dev).view(1, H, 1)
    d = depth[:, 0]
    P = torch.stack([(u - cx) / fx * d, (v - cy) / fy * d, d], dim=-1).view(B, H * W, 3)
    R, t = cam2world[:, :3, :3], cam2world[:, :3, 3]
    return P @ R.transpose(1, 2) + t.unsqueeze(1)


def stitch_point_cloud(depths: torch.Tensor, K: torch.Tensor, cam2world: torch.Tensor, *, stride: int = 1,
                       max_depth: float = MAX_DEPTH) -> torch.Tensor:
    """The reconstruction of README.md:29: every `stride`-th pixel of every frame, back-projected into the world frame of
    camera 0; pixels at or beyond `max_depth` are dropped.  depths [N,1,H,W], K [N,3,3], cam2world [N,4,4] -> [M,3]."""
    N, _, H, W = depths.shape
    pts = backproject(depths, K, cam2world).view(N, H, W, 3)[:, ::stride, ::stride]
    keep = depths[:, 0, ::stride, ::stride] < max_depth
    return pts[keep]


def make_models(seed: int = 0, dtype=torch.float32):
    dn, pn = DepthNet().to(dtype), PoseNet().to(dtype)
    init_weights(dn, seed)
    init_weights(pn, seed + 1)
    return dn, pn
