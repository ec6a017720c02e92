#!/usr/bin/env python3
"""Where does the host's enqueue time of a training step go?  cProfile over N steps (GPU work is asynchronous: the profile
shows the Python / C-ABI call path only).   python tools/host_profile.py [steps=200]"""
import cProfile
import os
import pstats
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from coivo_amd import functional as Fh, nn as hnn, synth  # noqa: E402
from coivo_amd.optim import FusedAdam  # noqa: E402


def main():
    B, H, W, N = 8, 256, 320, int(sys.argv[1]) if len(sys.argv) > 1 else 200
    dev = torch.device("cuda:0")
    dn, pn = hnn.DepthNet(compute_dtype=torch.bfloat16, device=dev), hnn.PoseNet(compute_dtype=torch.bfloat16, device=dev)
    opt = FusedAdam([dn, pn], lr=1e-4)
    batch = synth.make_batch(B, H, W, seed=1, device=dev)
    frames = torch.cat([batch["tgt"], batch["ref"]]).contiguous()
    K = batch["K"]
    one = torch.ones((), device=dev)

    def step():
        opt.zero_grad()
        d_t, d_r, d_l = dn.forward_pair_split(frames)
        tgt, ref = frames[:B], frames[B:]
        pose, a, b = pn(tgt, ref, d_t, d_r)
        loss = Fh.photometric_loss(tgt, ref, d_l, pose, K, a, b)
        loss.backward(gradient=one)
        opt.step()

    for _ in range(20):
        step()
    torch.cuda.synchronize()
    pr = cProfile.Profile()
    t0 = time.perf_counter()
    pr.enable()
    for _ in range(N):
        step()
    pr.disable()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    print(f"host enqueue {1e3 * (t1 - t0) / N:.3f} ms/step under cProfile")
    st = pstats.Stats(pr)
    st.sort_stats("cumulative").print_stats(28)
    st.sort_stats("tottime").print_stats(18)


if __name__ == "__main__":
    main()
