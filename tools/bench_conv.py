#!/usr/bin/env python3
"""Per-layer timing of the conv kernels (forward / dgrad / wgrad) on the DepthNet + PoseNet layer shapes at B images.
   python tools/bench_conv.py [B=16] [dtype=bf16] [fwdonly|wgradonly]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from coivo_amd import ops  # noqa: E402

ENC = (32, 64, 128, 256, 512)
DEC = (16, 32, 64, 128, 256)


def layers(B, H, W):
    out = []
    h, w, cin = H, W, 8
    for i, c in enumerate(ENC, 1):
        out.append((f"enc{i}a", B, h, w, cin, 0, False, c, 2)); h //= 2; w //= 2
        out.append((f"enc{i}b", B, h, w, c, 0, False, c, 1)); cin = c
    for i in range(5, 0, -1):
        d = DEC[i - 1]; h *= 2; w *= 2
        out.append((f"up{i}", B, h, w, cin, 0, True, d, 1))
        skip = ENC[i - 2] if i >= 2 else 0
        out.append((f"iconv{i}", B, h, w, d, skip, False, d, 1)); cin = d
    return out


def timeit(fn, n=int(os.environ.get("CONV_BENCH_ITERS", "30"))):
    for _ in range(5 if n >= 10 else 1):
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
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    dt = torch.float32 if (len(sys.argv) > 2 and sys.argv[2] == "f32") else torch.bfloat16
    fwdonly = len(sys.argv) > 3 and sys.argv[3] == "fwdonly"
    wonly = len(sys.argv) > 3 and sys.argv[3] == "wgradonly"          # (the other two columns then read 1.0)
    dev = torch.device("cuda:0")
    tot = [0.0, 0.0, 0.0]
    gsum = [0.0, 0.0, 0.0]
    print(f"{'layer':8s} {'shape':28s} {'GFLOP':>7s} | {'fwd us':>8s} {'GF/us':>6s} | {'dgrad us':>8s} {'GF/us':>6s} | {'wgrad us':>8s} {'GF/us':>6s}")
    HH, WW = (int(v) for v in os.environ.get("CONV_BENCH_HW", "256x320").split("x"))      # e.g. 512x640 = BASELINE configs[2]
    for (name, b, hi, wi, c0, c1, up, cout, stride) in layers(B, HH, WW):
        d = ops.conv_desc(dt, b, hi, wi, c0, cout, stride=stride, C1=c1, up0=up)
        hs, ws = (hi // 2, wi // 2) if up else (hi, wi)
        x0 = torch.randn(b, hs, ws, c0, device=dev).relu().to(dt)
        x1 = torch.randn(b, hi, wi, c1, device=dev).relu().to(dt) if c1 else None
        cin = c0 + c1
        wf = (torch.randn(cout, 9, cin, device=dev) * 0.05).to(dt)
        wb = (torch.randn(cin, 9, cout, device=dev) * 0.05).to(dt)
        bias = torch.zeros(cout, device=dev)
        y = torch.empty(b, d.Ho, d.Wo, cout, device=dev, dtype=dt)
        dy = torch.randn(b, d.Ho, d.Wo, cout, device=dev).to(dt)
        dx = torch.empty_like(x0)
        dw = torch.zeros(cout, 9, cin, device=dev)
        db = torch.zeros(cout, device=dev)
        gf = 2.0 * cout * cin * 9 * d.Ho * d.Wo * b / 1e9
        tf = 1.0 if wonly else timeit(lambda: ops.conv_fwd(d, x0, x1, wf, bias, y))
        td = 1.0 if (fwdonly or wonly) else timeit(lambda: ops.conv_dgrad(d, 0, dy, wb, x0, dx, False))
        # CONV_BENCH_DET=1: the deterministic form (per-split slabs + the per-layer reduce launch) instead of float atomics
        scr = ops.conv_wgrad_scratch(d, dev) if os.environ.get("CONV_BENCH_DET") else None
        tw = 1.0 if fwdonly else timeit(lambda: ops.conv_wgrad(d, x0, x1, dy, dw, db, scr))
        gd = gf * c0 / cin
        tot[0] += tf; tot[1] += td; tot[2] += tw
        gsum[0] += gf; gsum[1] += gd; gsum[2] += gf
        shape = "%d+%d->%d @%dx%d s%d%s" % (c0, c1, cout, d.Ho, d.Wo, stride, " up" if up else "")
        print(f"{name:8s} {shape:28s} {gf:7.2f} | {tf:8.1f} {gf / tf:6.1f} | {td:8.1f} {gd / td:6.1f} | {tw:8.1f} {gf / tw:6.1f}")
    print(f"totals: fwd {tot[0]:.0f} us  dgrad {tot[1]:.0f} us  wgrad {tot[2]:.0f} us")
    if not fwdonly:
        # 1 GFLOP/us = 1000 TFLOP/s; dgrad of the first layer (gradient w.r.t. the images) is not computed
        peak = 2500.0 if dt == torch.bfloat16 else 2500.0 / 16      # dense bf16 16x16x32 / f32 16x16x4 MFMA, TFLOP/s
        print("MFMA rate per pass (of %.0f TFLOP/s dense): fwd %.1f %%  dgrad %.1f %%  wgrad %.1f %%" %
              ((peak,) + tuple(100.0 * g / t * 1e3 / peak for g, t in zip(gsum, tot))))


if __name__ == "__main__":
    main()
