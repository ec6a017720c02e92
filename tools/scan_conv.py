#!/usr/bin/env python3
"""Scaling scan of the conv forward kernel: time vs C_in at fixed output, and vs batch (intercept / slope)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from coivo_amd import ops


def t_conv(B, H, W, C0, Cout, n=40):
    dev = torch.device("cuda:0"); dt = torch.bfloat16
    d = ops.conv_desc(dt, B, H, W, C0, Cout)
    x = torch.randn(B, H, W, C0, device=dev).relu().to(dt)
    w = (torch.randn(Cout, 9, C0, device=dev) * 0.05).to(dt)
    bias = torch.zeros(Cout, device=dev)
    y = torch.empty(B, H, W, Cout, device=dev, dtype=dt)
    for _ in range(5):
        ops.conv_fwd(d, x, None, w, bias, y)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        ops.conv_fwd(d, x, None, w, bias, y)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for (B, H, W, Cout) in ((16, 8, 10, 512), (16, 8, 10, 64), (1, 8, 10, 64), (16, 16, 20, 256), (16, 64, 80, 64)):
    print(f"B={B} {H}x{W} Cout={Cout}: " + "  ".join(f"C{c}:{t_conv(B, H, W, c, Cout):6.1f}us" for c in (32, 64, 128, 256, 512)))
