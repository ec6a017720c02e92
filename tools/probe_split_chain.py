#!/usr/bin/env python3
"""Would two half-batch DepthNet forward chains on two streams beat one full-batch chain?  (The chain alternates between
HBM-bound full-resolution layers and latency-bound deep layers at ~1 wave per SIMD: two staggered chains are complementary.)
   python tools/probe_split_chain.py [images=16]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from coivo_amd import nn as hnn  # noqa: E402


def timeit(fn, n=30):
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


def main():
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    dev = torch.device("cuda:0")
    dn = hnn.DepthNet(compute_dtype=torch.bfloat16, device=dev)
    with torch.no_grad():
        for name, p in dn.named_parameters():
            if name.endswith("weight"):
                p.copy_(torch.randn(p.shape, device=dev) * (2.0 / (p.shape[1] * 9)) ** 0.5)
    x = torch.rand(N, 3, 256, 320, device=dev)
    parts = [2, 4]
    streams = [torch.cuda.Stream() for _ in range(max(parts))]
    main_s = torch.cuda.current_stream()

    def whole():
        with torch.no_grad():
            dn(x)

    def split(k):
        def run():
            e = torch.cuda.Event()
            e.record(main_s)
            with torch.no_grad():
                for i in range(k):
                    streams[i].wait_event(e)
                    with torch.cuda.stream(streams[i]):
                        dn(x[i * N // k:(i + 1) * N // k])
            for i in range(k):
                main_s.wait_stream(streams[i])
        return run

    print(f"DepthNet forward, {N} images 256x320 bf16: one chain {timeit(whole):.1f} us")
    for k in parts:
        print(f"   {k} chains of {N // k} images on {k} streams: {timeit(split(k)):.1f} us")
    xs = x[:N // 2].contiguous()
    print(f"   (one chain of {N // 2} images alone: {timeit(lambda: dn(xs)):.1f} us)")


if __name__ == "__main__":
    main()
