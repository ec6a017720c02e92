#!/usr/bin/env python3
"""Time the fused warp/LCC/SSIM/L1 forward + backward alone (HIP events), at BASELINE configs[2] by default.

    python tools/bench_loss.py [B H W]
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from coivo_amd import functional as Fh  # noqa: E402
from coivo_amd import synth  # noqa: E402


def run(B, H, W, iters=30):
    dev = torch.device("cuda:0")
    nb = min(B, 4)
    b = synth.make_batch(nb, H, W, seed=77, device=dev)
    rep = lambda t: t.repeat((B + nb - 1) // nb, *([1] * (t.dim() - 1)))[:B].contiguous()
    tgt, ref, K = rep(b["tgt"]), rep(b["ref"]), rep(b["K"])
    leaves = [rep(b[k]).requires_grad_(True) for k in ("gt_depth", "gt_pose", "gt_a", "gt_b")]
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    tf, tb = [], []
    for it in range(iters):
        ev[0].record()
        loss = Fh.photometric_loss(tgt, ref, leaves[0], leaves[1], K, leaves[2], leaves[3])
        ev[1].record()
        torch.autograd.grad(loss, leaves)
        ev[2].record()
        torch.cuda.synchronize()
        if it >= 5:
            tf.append(ev[0].elapsed_time(ev[1]) * 1e3)
            tb.append(ev[1].elapsed_time(ev[2]) * 1e3)
    tf.sort(); tb.sort()
    f, bw = tf[len(tf) // 2], tb[len(tb) // 2]
    px = B * H * W
    print(f"B={B} {W}x{H}: fwd {f:8.1f} us ({28 * px / f / 1e3:7.1f} GB/s)  bwd {bw:8.1f} us ({32 * px / bw / 1e3:7.1f} GB/s)  "
          f"fwd+bwd {f + bw:8.1f} us = {60 * px / (f + bw) / 1e3:7.1f} GB/s = {60 * px / (f + bw) / 1e3 / 8000:.3f} of 8 TB/s  loss={loss.item():.6f}")


if __name__ == "__main__":
    if len(sys.argv) == 4:
        run(*map(int, sys.argv[1:]))
    else:
        run(32, 512, 640)
        run(8, 256, 320)
