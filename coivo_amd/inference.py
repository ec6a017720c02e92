"""Inference side of the path: depth maps + relative poses -> trajectory -> stitched point cloud.

Reference: README.md:9 ("complete 3D reconstruction of the intestine"), README.md:29 ("stitching together the dense depth
maps of each frame using the colonoscopic trajectory").  Spec: oracle/colvo_spec.py integrate_trajectory / backproject /
stitch_point_cloud (oracle/SPEC.md §6c).  The per-pixel work runs in csrc/reconstruct.hip; the trajectory integration is N
products of 4x4 matrices and is done on the host in float64 (it is control flow, not a kernel).
"""
from __future__ import annotations

import math
from typing import NamedTuple, Optional

import torch

from . import _lib

MAX_DEPTH = 10.0     # spec: MAX_DEPTH


def _chk(t: torch.Tensor, name: str, shape) -> torch.Tensor:
    if not isinstance(t, torch.Tensor) or not t.is_cuda or t.dtype != torch.float32 or tuple(t.shape) != tuple(shape):
        raise ValueError(f"{name}: expected a float32 CUDA tensor of shape {tuple(shape)}, got "
                         f"{getattr(t, 'dtype', None)} {tuple(getattr(t, 'shape', ()))} on {getattr(t, 'device', None)}")
    return t.contiguous()


def pose_to_matrix4(pose: torch.Tensor) -> torch.Tensor:
    """[N,6] (tx,ty,tz,rx,ry,rz; R = Rz Ry Rx, spec §4) -> [N,4,4] float64 on the CPU."""
    p = pose.detach().to("cpu", torch.float64)
    if p.dim() != 2 or p.shape[1] != 6:
        raise ValueError("pose_to_matrix4: expected [N,6]")
    out = torch.zeros(p.shape[0], 4, 4, dtype=torch.float64)
    for n in range(p.shape[0]):
        tx, ty, tz, rx, ry, rz = (float(v) for v in p[n])
        cx, sx, cy, sy, cz, sz = math.cos(rx), math.sin(rx), math.cos(ry), math.sin(ry), math.cos(rz), math.sin(rz)
        out[n] = torch.tensor([
            [cz * cy, cz * sy * sx - sz * cx, cz * sy * cx + sz * sx, tx],
            [sz * cy, sz * sy * sx + cz * cx, sz * sy * cx - cz * sx, ty],
            [-sy, cy * sx, cy * cx, tz],
            [0.0, 0.0, 0.0, 1.0]], dtype=torch.float64)
    return out


def integrate_trajectory(rel_poses: torch.Tensor) -> torch.Tensor:
    """Camera-to-world transforms of frames 0..N from the N relative poses of consecutive pairs (spec:
    integrate_trajectory): rel_poses[k] maps frame-k points into frame k+1; world = camera 0;
    M_0 = I, M_{k+1} = M_k @ inverse(T_k).  -> [N+1,4,4] float64 on the CPU."""
    T = pose_to_matrix4(rel_poses)
    M = [torch.eye(4, dtype=torch.float64)]
    for k in range(T.shape[0]):
        R, t = T[k, :3, :3], T[k, :3, 3]
        Tinv = torch.eye(4, dtype=torch.float64)
        Tinv[:3, :3] = R.t()
        Tinv[:3, 3] = -(R.t() @ t)
        M.append(M[-1] @ Tinv)
    return torch.stack(M)


def backproject(depth: torch.Tensor, K: torch.Tensor, cam2world: torch.Tensor) -> torch.Tensor:
    """depth [B,1,H,W], K [B,3,3], cam2world [B,4,4] -> world points [B,H*W,3] (spec: backproject)."""
    lib = _lib.load()
    if depth.dim() != 4:
        raise ValueError("backproject: depth must be [B,1,H,W]")
    B, _, H, W = depth.shape
    depth = _chk(depth, "depth", (B, 1, H, W))
    K = _chk(K, "K", (B, 3, 3))
    cam2world = _chk(cam2world, "cam2world", (B, 4, 4))
    points = torch.empty(B, H * W, 3, device=depth.device, dtype=torch.float32)
    _lib.check(lib.colvo_backproject(_lib.ptr(depth), _lib.ptr(K), _lib.ptr(cam2world), B, H, W, _lib.ptr(points),
                                     _lib.stream_ptr()), "colvo_backproject")
    return points


def stitch_point_cloud(depths: torch.Tensor, K: torch.Tensor, cam2world: torch.Tensor, *, stride: int = 1,
                       max_depth: float = MAX_DEPTH) -> torch.Tensor:
    """Every `stride`-th pixel of every frame with depth < max_depth, in the world frame, frame-major / row-major order
    (spec: stitch_point_cloud).  depths [N,1,H,W], K [N,3,3], cam2world [N,4,4] -> [M,3].  Reads the point count back
    (one 4-byte copy) to size the result."""
    lib = _lib.load()
    if depths.dim() != 4:
        raise ValueError("stitch_point_cloud: depths must be [N,1,H,W]")
    N, _, H, W = depths.shape
    if stride < 1:
        raise ValueError("stitch_point_cloud: stride must be >= 1")
    depths = _chk(depths, "depths", (N, 1, H, W))
    K = _chk(K, "K", (N, 3, 3))
    cam2world = _chk(cam2world, "cam2world", (N, 4, 4))
    cap = N * -(-H // stride) * -(-W // stride)
    ws = torch.empty(int(lib.colvo_stitch_workspace_ints(N, H, W, stride)), device=depths.device, dtype=torch.int32)
    count = torch.empty(1, device=depths.device, dtype=torch.int32)
    points = torch.empty(cap, 3, device=depths.device, dtype=torch.float32)
    _lib.check(lib.colvo_stitch_point_cloud(_lib.ptr(depths), _lib.ptr(K), _lib.ptr(cam2world), N, H, W, stride,
                                            float(max_depth), _lib.ptr(ws), _lib.ptr(points), _lib.ptr(count),
                                            _lib.stream_ptr()), "colvo_stitch_point_cloud")
    return points[: int(count.item())]


class Reconstruction(NamedTuple):
    depths: torch.Tensor        # [N+1,1,H,W]
    rel_poses: torch.Tensor     # [N,6]    frame k -> frame k+1
    cam2world: torch.Tensor     # [N+1,4,4] float64, CPU
    points: torch.Tensor        # [M,3]    world frame (camera 0)


@torch.no_grad()
def reconstruct_sequence(depth_net, pose_net, frames: torch.Tensor, K: torch.Tensor, *, stride: int = 4,
                         max_depth: float = MAX_DEPTH, chunk: int = 16) -> Reconstruction:
    """frames [N+1,3,H,W] of one sequence, K [3,3] or [N+1,3,3] -> depth of every frame, the pose of every consecutive
    pair (DCDP: PoseNet sees both depth maps), the integrated trajectory and the stitched cloud."""
    n = frames.shape[0]
    if n < 2:
        raise ValueError("reconstruct_sequence: need at least two frames")
    if K.dim() == 2:
        K = K.unsqueeze(0).expand(n, 3, 3)
    K = K.to(frames.device, torch.float32).contiguous()
    depths = torch.cat([depth_net(frames[i:i + chunk].contiguous()) for i in range(0, n, chunk)])
    poses = []
    for i in range(0, n - 1, chunk):
        j = min(i + chunk, n - 1)
        pose, _, _ = pose_net(frames[i:j].contiguous(), frames[i + 1:j + 1].contiguous(),
                              depths[i:j].contiguous(), depths[i + 1:j + 1].contiguous())
        poses.append(pose)
    rel = torch.cat(poses)
    traj = integrate_trajectory(rel)
    cloud = stitch_point_cloud(depths, K, traj.to(frames.device, torch.float32), stride=stride, max_depth=max_depth)
    return Reconstruction(depths, rel, traj, cloud)
