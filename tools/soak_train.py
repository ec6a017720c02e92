#!/usr/bin/env python3
"""Soak: N training steps on ONE synthetic batch in bf16 and in f32 mode; prints the loss every 50 steps.  Both curves must
fall and stay close (bf16 conv / fp32 loss vs exact-f32): a cheap end-to-end check against silent corruption.
   python tools/soak_train.py [steps=300]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from coivo_amd import functional as Fh, nn as hnn, synth  # noqa: E402
from coivo_amd.optim import FusedAdam  # noqa: E402


def run(dtype, steps, B=4, H=128, W=160):
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    dn, pn = hnn.DepthNet(compute_dtype=dtype, device=dev), hnn.PoseNet(compute_dtype=dtype, device=dev)
    g = torch.Generator().manual_seed(0)
    with torch.no_grad():
        for net in (dn, pn):
            for name, p in net.named_parameters():
                if name.endswith("weight"):
                    fan_in = p.shape[1] * p.shape[2] * p.shape[3]
                    p.copy_((torch.randn(p.shape, generator=g) * (2.0 / fan_in) ** 0.5).to(dev))
    opt = FusedAdam([dn, pn], lr=1e-4)
    b = synth.make_batch(B, H, W, seed=7, device=dev)
    frames = torch.cat([b["tgt"], b["ref"]]).contiguous()
    out = []
    for it in range(steps + 1):
        opt.zero_grad()
        d_t, d_r, d_l = dn.forward_pair_split(frames)
        pose, a, bb = pn(frames[:B], frames[B:], d_t, d_r)
        loss = Fh.photometric_loss(frames[:B], frames[B:], d_l, pose, b["K"], a, bb)
        loss.backward()
        opt.step()
        if it % 50 == 0:
            out.append(loss.item())
    return out


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    lb = run(torch.bfloat16, n)
    lf = run(torch.float32, n)
    print("bf16:", " ".join(f"{v:.5f}" for v in lb))
    print("f32 :", " ".join(f"{v:.5f}" for v in lf))
    ok = all(v == v for v in lb + lf) and lb[-1] < lb[0] and lf[-1] < lf[0] and abs(lb[-1] - lf[-1]) < 0.1 * lf[0]
    print("OK" if ok else "MISMATCH")
    sys.exit(0 if ok else 1)
