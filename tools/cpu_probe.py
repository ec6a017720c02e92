#!/usr/bin/env python3
"""Is the eager training step bound by the host's enqueue path?  Times N steps twice: the wall time until the LAST launch
has been enqueued (no synchronisation) and until the GPU has drained.  enqueue ~= total  =>  host-bound."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from coivo_amd import functional as Fh, nn as hnn, synth  # noqa: E402
from coivo_amd.optim import FusedAdam  # noqa: E402


def main():
    B, H, W, N = 8, 256, 320, int(sys.argv[1]) if len(sys.argv) > 1 else 40
    dev = torch.device("cuda:0")
    dn, pn = hnn.DepthNet(compute_dtype=torch.bfloat16, device=dev), hnn.PoseNet(compute_dtype=torch.bfloat16, device=dev)
    opt = FusedAdam([dn, pn], lr=1e-4)
    batch = synth.make_batch(B, H, W, seed=1, device=dev)
    frames = torch.cat([batch["tgt"], batch["ref"]]).contiguous()
    K = batch["K"]

    def step():
        opt.zero_grad()
        d_t, d_r, d_l = dn.forward_pair_split(frames)
        tgt, ref = frames[:B], frames[B:]
        pose, a, b = pn(tgt, ref, d_t, d_r)
        loss = Fh.photometric_loss(tgt, ref, d_l, pose, K, a, b)
        loss.backward()
        opt.step()

    for _ in range(10):
        step()
    torch.cuda.synchronize()
    for rep in range(3):
        t0 = time.perf_counter()
        for _ in range(N):
            step()
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        print(f"rep {rep}: enqueue {1e3 * (t1 - t0) / N:.3f} ms/step   total {1e3 * (t2 - t0) / N:.3f} ms/step")
    if len(sys.argv) > 2 and sys.argv[2] == "profile":          # where the host time goes (cProfile, cumulative, top 45)
        import cProfile
        import pstats
        pr = cProfile.Profile()
        pr.enable()
        for _ in range(N):
            step()
        pr.disable()
        torch.cuda.synchronize()
        pstats.Stats(pr).sort_stats("cumulative").print_stats(45)
        pstats.Stats(pr).sort_stats("tottime").print_stats(25)
        # the backward bodies run in autograd's device thread, which the profile above does not see: profile them there
        prb = cProfile.Profile()
        for net in (dn, pn):
            orig = net._backward_impl

            def wrapped(*a, _orig=orig, **kw):
                prb.enable()
                try:
                    return _orig(*a, **kw)
                finally:
                    prb.disable()
            net._backward_impl = wrapped
        t0 = time.perf_counter()
        for _ in range(N):
            step()
        torch.cuda.synchronize()
        print(f"backward bodies (autograd thread), {N} steps:")
        pstats.Stats(prb).sort_stats("cumulative").print_stats(30)


if __name__ == "__main__":
    main()
