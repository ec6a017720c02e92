#!/usr/bin/env python3
"""Host time of the phases of the first steps after a device synchronisation (the driver protocol times 20 steps from a drained GPU;
its first step reads 1.8-2.6 ms where the steady state is 1.27):   python tools/first_step_probe.py [rounds=4]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from coivo_amd import functional as Fh, nn as hnn, synth  # noqa: E402
from coivo_amd.optim import FusedAdam  # noqa: E402


def main():
    B, H, W = 8, 256, 320
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    dev = torch.device("cuda:0")
    dn, pn = hnn.DepthNet(compute_dtype=torch.bfloat16, device=dev), hnn.PoseNet(compute_dtype=torch.bfloat16, device=dev)
    opt = FusedAdam([dn, pn], lr=1e-4, zero_grad_in_step=True)
    batch = synth.make_batch(B, H, W, seed=1, device=dev)
    frames = torch.cat([batch["tgt"], batch["ref"]]).contiguous()
    K = batch["K"]
    one = torch.ones((), device=dev)
    tgt, ref = frames[:B], frames[B:]
    pc = time.perf_counter

    def step(rec):
        t0 = pc()
        opt.zero_grad()
        d_t, d_r, d_l = dn.forward_pair_split(frames)
        t1 = pc()
        pose, a, b = pn(tgt, ref, d_t, d_r)
        loss = Fh.photometric_loss(tgt, ref, d_l, pose, K, a, b)
        t2 = pc()
        loss.backward(gradient=one)
        t3 = pc()
        opt.step()
        t4 = pc()
        rec.append((t1 - t0, t2 - t1, t3 - t2, t4 - t3))

    for _ in range(25):
        step([])
    for r in range(rounds):
        torch.cuda.synchronize()
        what = ("sync only", "sync + 20 ms idle", "sync + gc.collect()", "sync + gc.collect() + gc.disable()")[r % 4]
        if r % 4 == 1:
            time.sleep(0.02)
        if r % 4 >= 2:
            import gc
            gc.collect()
            if r % 4 == 3:
                gc.disable()
        print(f"round {r}: {what}")
        rec = []
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(7)]
        ev[0].record()
        for i in range(6):
            step(rec)
            ev[i + 1].record()
        torch.cuda.synchronize()
        import gc as _gc
        _gc.enable()
        for i, (a, b, c, d) in enumerate(rec):
            print(f"round {r} step {i}: host depthnet-fwd {a * 1e3:.3f}  posenet+loss {b * 1e3:.3f}  backward {c * 1e3:.3f}  adam {d * 1e3:.3f}  "
                  f"= {(a + b + c + d) * 1e3:.3f} ms | gpu {ev[i].elapsed_time(ev[i + 1]):.3f} ms")


if __name__ == "__main__":
    main()
