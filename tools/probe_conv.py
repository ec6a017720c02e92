#!/usr/bin/env python3
"""Run ONE conv layer shape in a loop (for rocprofv3 --pmc):  python tools/probe_conv.py B H W C0 Cout [stride] [iters]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from coivo_amd import ops  # noqa: E402

B, H, W, C0, Cout = map(int, sys.argv[1:6])
stride = int(sys.argv[6]) if len(sys.argv) > 6 else 1
iters = int(sys.argv[7]) if len(sys.argv) > 7 else 20
dev = torch.device("cuda:0")
dt = torch.bfloat16
d = ops.conv_desc(dt, B, H, W, C0, Cout, stride=stride)
x = torch.randn(B, H, W, C0, device=dev).relu().to(dt)
w = (torch.randn(Cout, 9, C0, device=dev) * 0.05).to(dt)
bias = torch.zeros(Cout, device=dev)
y = torch.empty(B, d.Ho, d.Wo, Cout, device=dev, dtype=dt)
for _ in range(iters):
    ops.conv_fwd(d, x, None, w, bias, y)
torch.cuda.synchronize()
print("done")
