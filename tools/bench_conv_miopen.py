#!/usr/bin/env python3
"""Vendor-library yardstick for tools/bench_conv.py's per-layer table: the same 20 DepthNet conv layers through
torch.nn.functional.conv2d / aten.convolution_backward (= MIOpen on ROCm), bf16 channels_last, each pass timed alone.
Tools only -- nothing here is imported by the product or runs inside bench.py's timed region.

   python tools/bench_conv_miopen.py [B=16] [out.csv]      (CONV_BENCH_HW=512x640 for the size of configs[2];
                                                             CONV_MIOPEN_FIND=1: MIOpen's benchmark search instead of its immediate mode)

What is timed is the bare convolution on a MATERIALISED input: the nearest-2x up-sampling and the skip concat that the hand kernels
fold into their gather are given to MIOpen for free (its input tensor already holds them), and its input gradient of an up-sampled
layer stops at the up-sampled tensor (no 2x2 sum-pool).  The column is therefore a lower bound of what the library path would cost.
Rows are appended to the CSV as they are measured, so a run cut short by its time limit still leaves a table."""
import os
import sys
import time

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from bench_conv import layers  # noqa: E402  (layer list only; importing it loads libcolvo, which is not used here)


def timeit(fn, n=int(os.environ.get("CONV_BENCH_ITERS", "30"))):
    for _ in range(3):
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
    out = sys.argv[2] if len(sys.argv) > 2 else None
    torch.backends.cudnn.benchmark = bool(int(os.environ.get("CONV_MIOPEN_FIND", "0")))
    dev = torch.device("cuda:0")
    HH, WW = (int(v) for v in os.environ.get("CONV_BENCH_HW", "256x320").split("x"))
    f = open(out, "w") if out else sys.stdout
    f.write("# torch %s, MIOpen through aten (cudnn.benchmark=%s), bf16 channels_last, B=%d frames of %dx%d; us per call\n"
            % (torch.__version__, torch.backends.cudnn.benchmark, B, HH, WW))
    f.write("layer,gflop,miopen_fwd_us,miopen_dgrad_us,miopen_wgrad_us\n")
    f.flush()
    tot = [0.0, 0.0, 0.0]
    t_start = time.time()
    for (name, b, hi, wi, c0, c1, up, cout, stride) in layers(B, HH, WW):
        cin = c0 + c1
        x = torch.randn(b, cin, hi, wi, device=dev).relu().to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
        w = (torch.randn(cout, cin, 3, 3, device=dev) * 0.05).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
        ho, wo = (hi - 1) // stride + 1, (wi - 1) // stride + 1
        dy = torch.randn(b, cout, ho, wo, device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
        gf = 2.0 * cout * cin * 9 * ho * wo * b / 1e9

        def bwd(mask):
            return torch.ops.aten.convolution_backward(dy, x, w, None, [stride, stride], [1, 1], [1, 1], False, [0, 0], 1, mask)

        tf = timeit(lambda: F.conv2d(x, w, None, stride, 1))
        td = timeit(lambda: bwd([True, False, False]))
        tw = timeit(lambda: bwd([False, True, False]))
        tot = [tot[0] + tf, tot[1] + td, tot[2] + tw]
        f.write(f"{name},{gf:.2f},{tf:.1f},{td:.1f},{tw:.1f}\n")
        f.flush()
        print(f"{name:8s} {gf:7.2f} GFLOP | miopen fwd {tf:8.1f} dgrad {td:8.1f} wgrad {tw:8.1f} us   (+{time.time() - t_start:.0f} s)", flush=True)
        del x, w, dy
    f.write(f"# totals: fwd {tot[0]:.0f} us  dgrad {tot[1]:.0f} us  wgrad {tot[2]:.0f} us\n")
    f.flush()


if __name__ == "__main__":
    main()
