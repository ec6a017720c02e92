#!/usr/bin/env python3
"""The two fused full-resolution kernels alone (GPU box): k_fwd16_head (colvo_conv_head_fused) and k_bwd16 in the HEAD form
(colvo_conv_bwd_fused with the depth head's d(pre)), at B frames of 256x320 -- hip-event time per launch and the rate against the
tensors each must move (algorithmic bytes: fwd = x in, y + depth out; bwd = y, x, d(pre) in, dx out).

    python tools/bench_narrow.py [frames=64] [launches=30]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from coivo_amd import ops  # noqa: E402


def timed(fn, n):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
    ev[0].record()
    for i in range(n):
        fn()
        ev[i + 1].record()
    torch.cuda.synchronize()
    per = sorted(ev[i].elapsed_time(ev[i + 1]) * 1e3 for i in range(n))
    return per[len(per) // 2]


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 30
    H, W = 256, 320
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(7)
    dt = torch.bfloat16
    d = ops.conv_desc(dt, B, H, W, 16, 16)
    x = torch.randn(B, H, W, 16, generator=g).relu().to(dev).to(dt)
    w = (torch.randn(16, 9, 16, generator=g) * 0.15).to(dev).to(dt)
    bias = (torch.randn(16, generator=g) * 0.1).to(dev)
    wh = (torch.randn(1, 9, 16, generator=g) * 0.2).to(dev)
    bh = torch.tensor([0.3], device=dev)
    y = torch.empty(B, H, W, 16, device=dev, dtype=dt)
    depth = torch.empty(B, 1, H, W, device=dev)
    pose_in = torch.zeros(B // 2, H, W, 8, device=dev, dtype=dt) if B % 2 == 0 and not os.environ.get("NARROW_NO_POSE") else None
    px = B * H * W
    t = timed(lambda: ops.conv_head_fused(d, x, w, bias, wh, bh, y, depth, pose_in), n)
    by = px * (32 + 32 + 4 + (2 if pose_in is not None else 0))
    print(f"k_fwd16_head      B={B}: {t:8.1f} us   {by / 1e6:7.1f} MB algorithmic  {by / t / 1e6:6.2f} TB/s")
    dpre = torch.randn(B, H, W, generator=g).to(dev)
    w_bwd = (torch.randn(16, 9, 16, generator=g) * 0.1).to(dev).to(dt)
    dx = torch.empty_like(x)
    dw, db = torch.zeros(16, 9, 16, device=dev), torch.zeros(16, device=dev)
    t = timed(lambda: ops.conv_bwd_fused(d, y, w_bwd, x, True, dx, dw, db, dpre, wh), n)
    by = px * (32 + 32 + 4 + 32)
    print(f"k_bwd16<HEAD>     B={B}: {t:8.1f} us   {by / 1e6:7.1f} MB algorithmic  {by / t / 1e6:6.2f} TB/s")
    t = timed(lambda: ops.conv_bwd_fused(d, y, w_bwd, x, True, dx, dw, db), n)
    by = px * (32 + 32 + 32)
    print(f"k_bwd16<plain>    B={B}: {t:8.1f} us   {by / 1e6:7.1f} MB algorithmic  {by / t / 1e6:6.2f} TB/s")


if __name__ == "__main__":
    main()
