"""Probe: DepthNet's forward pass on 16 frames as one chain against two 8-frame chains on two streams (and four 4-frame ones)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from coivo_amd import nn as hnn

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
dt = torch.bfloat16
torch.manual_seed(0)
x = torch.rand(B, 3, 256, 320, device="cuda")
nets = [hnn.DepthNet(dt) for _ in range(5)]
for n in nets[1:]:
    n.load_state_dict(nets[0].state_dict())
cur = torch.cuda.current_stream()
streams = [torch.cuda.Stream() for _ in range(4)]

def one():
    return nets[0](x)

def split(k):
    def f():
        outs = []
        h = B // k
        for i in range(k):
            s = streams[i]
            s.wait_stream(cur)
            with torch.cuda.stream(s):
                outs.append(nets[1 + i](x[i * h:(i + 1) * h]))
        for i in range(k):
            cur.wait_stream(streams[i])
        return outs
    return f

def timeit(fn, n=60, warm=15):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / n * 1e3)
    return sorted(ts)[len(ts) // 2]

with torch.no_grad():
    a = one(); b = split(2)()
    torch.cuda.synchronize()
    print("max |d| one vs two chains:", (a - torch.cat(b)).abs().max().item())
    for rep in range(2):
        print(f"B={B}: one chain {timeit(one):.1f} us   two chains {timeit(split(2)):.1f} us   four chains {timeit(split(4)):.1f} us")
