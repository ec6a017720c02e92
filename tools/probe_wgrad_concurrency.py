#!/usr/bin/env python3
"""What would ONE grouped weight-gradient launch buy?  The 20 weight-gradient kernels of DepthNet are independent of each
other; in the training step each runs at about one 4-wave workgroup per CU.  This probe enqueues all of them with no
dependencies between them on S streams (S = 1: back to back, the isolated sum; S > 1: up to S kernels share the CUs) and
times the whole set -- an upper bound of what a single launch over the union of their grids could reach.
   python tools/probe_wgrad_concurrency.py [B=16] [streams=1,2,4,8] [pass=wgrad|fwd|dgrad]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from coivo_amd import ops  # noqa: E402
from tools.bench_conv import layers  # noqa: E402


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    ss = [int(v) for v in (sys.argv[2] if len(sys.argv) > 2 else "1,2,4,8").split(",")]
    which = sys.argv[3] if len(sys.argv) > 3 else "wgrad"
    dt = torch.bfloat16
    dev = torch.device("cuda:0")
    calls = []
    for (name, b, hi, wi, c0, c1, up, cout, stride) in layers(B, 256, 320):
        d = ops.conv_desc(dt, b, hi, wi, c0, cout, stride=stride, C1=c1, up0=up)
        hs, ws = (hi // 2, wi // 2) if up else (hi, wi)
        x0 = torch.randn(b, hs, ws, c0, device=dev).relu().to(dt)
        x1 = torch.randn(b, hi, wi, c1, device=dev).relu().to(dt) if c1 else None
        cin = c0 + c1
        dy = torch.randn(b, d.Ho, d.Wo, cout, device=dev).to(dt)
        dw = torch.zeros(cout, 9, cin, device=dev)
        db = torch.zeros(cout, device=dev)
        wf = (torch.randn(cout, 9, cin, device=dev) * 0.05).to(dt)
        wb = (torch.randn(cin, 9, cout, device=dev) * 0.05).to(dt)
        bias = torch.zeros(cout, device=dev)
        y = torch.empty(b, d.Ho, d.Wo, cout, device=dev, dtype=dt)
        dx = torch.empty_like(x0)
        if which == "wgrad":
            calls.append(lambda d=d, x0=x0, x1=x1, dy=dy, dw=dw, db=db: ops.conv_wgrad(d, x0, x1, dy, dw, db))
        elif which == "fwd":
            calls.append(lambda d=d, x0=x0, x1=x1, wf=wf, bias=bias, y=y: ops.conv_fwd(d, x0, x1, wf, bias, y))
        else:
            calls.append(lambda d=d, dy=dy, wb=wb, x0=x0, dx=dx: ops.conv_dgrad(d, 0, dy, wb, x0, dx, False))
    torch.cuda.synchronize()
    for S in ss:
        streams = [torch.cuda.Stream() for _ in range(S)]
        main_s = torch.cuda.current_stream()

        def run():
            e = torch.cuda.Event()
            e.record(main_s)
            for st in streams:
                st.wait_event(e)
            for i, c in enumerate(calls):
                with torch.cuda.stream(streams[i % S]):
                    c()
            for st in streams:
                main_s.wait_stream(st)
        for _ in range(3):
            run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 10
        e0.record()
        for _ in range(n):
            run()
        e1.record()
        torch.cuda.synchronize()
        print(f"{which}: {len(calls)} kernels on {S} stream(s): {e0.elapsed_time(e1) / n * 1e3:8.1f} us per set", flush=True)


if __name__ == "__main__":
    main()
