#!/usr/bin/env python3
"""Time the depth-head kernels alone (B=16 images, 256x320, 16 channels): python tools/bench_head.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from coivo_amd import ops  # noqa: E402

d = torch.device("cuda:0")
B, H, W, C = 16, 256, 320, 16
x = torch.randn(B, H, W, C, device=d).relu().to(torch.bfloat16)
w = torch.randn(1, 9, C, device=d) * 0.1
bias = torch.zeros(1, device=d)
depth = torch.empty(B, 1, H, W, device=d)
dd = torch.randn(B, 1, H, W, device=d)
scratch = torch.empty(B * H * W, device=d)
dx = torch.empty_like(x)
dw = torch.zeros(1, 9, C, device=d)
db = torch.zeros(1, device=d)


def t(fn, n=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


print("fwd        %.1f us" % t(lambda: ops.depth_head_fwd(x, w, bias, depth)))
print("dpre+dgrad %.1f us" % t(lambda: ops.depth_head_bwd(x, w, depth, dd, scratch, dx, None, None)))
print("wgrad      %.1f us" % t(lambda: ops.depth_head_wgrad(x, scratch, dw, db)))
