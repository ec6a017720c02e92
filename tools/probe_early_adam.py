#!/usr/bin/env python3
"""Feasibility probe (timing only, results are NOT valid in the `early` mode): what would the step gain if the Adam pass over the
layers whose gradients are final early ran on an idle stream WHILE the backward pass finishes?

    python3 tools/probe_early_adam.py [--pairs 8] [--steps 60] [--layer enc4a]

`early` launches the WHOLE update on PoseNet's (idle) weight-gradient stream behind an event recorded on the main stream when the
backward pass reaches --layer, skipping the joins: an upper bound of what a split update (early layers there, the rest at the end)
can hide.  Prints hip-event medians per step for base / early, alternating blocks."""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--pairs", type=int, default=8)
    ap.add_argument("--steps", type=int, default=60)
    ap.add_argument("--layer", default="enc4a")
    args = ap.parse_args()
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
    from coivo_amd import build
    build.ensure()
    from coivo_amd import functional as Fh, nn as hnn, synth
    from coivo_amd.optim import FusedAdam
    dev = torch.device("cuda:0")
    B, H, W = args.pairs, 256, 320
    dn, pn = hnn.DepthNet(compute_dtype=torch.bfloat16, device=dev), hnn.PoseNet(compute_dtype=torch.bfloat16, device=dev)
    g = torch.Generator().manual_seed(0)
    with torch.no_grad():
        for net in (dn, pn):
            for name, p in net.named_parameters():
                if name.endswith("weight"):
                    fan_in = p.shape[1] * p.shape[2] * p.shape[3]
                    p.copy_((torch.randn(p.shape, generator=g) * (2.0 / fan_in) ** 0.5).to(dev))
    opt = FusedAdam([dn, pn], lr=1e-4, zero_grad_in_step=True)
    b = synth.make_batch(B, H, W, seed=1234, device=dev)
    frames = torch.cat([b["tgt"], b["ref"]])
    tgt, ref, K = frames[:B], frames[B:], b["K"]
    one = torch.ones((), device=dev)
    main = torch.cuda.current_stream()
    ev = torch.cuda.Event()
    span = getattr(dn, args.layer).span

    def hook(m, lo, hi):
        if (lo, hi) == tuple(span):
            ev.record(main)
            return True
        return False

    def step_next():
        """timing probe of the OTHER overlap: the update on a side stream under the NEXT step's forward pass (the main stream does
        not wait for it at all here -- an upper bound; a real form lets the first layers' update run on the main stream and makes
        the forward pass wait for the rest where it first needs it)"""
        opt.zero_grad()
        d_t, d_r, d_l = dn.forward_pair_split(frames)
        pose, a, bb = pn(tgt, ref, d_t, d_r)
        loss = Fh.photometric_loss(tgt, ref, d_l, pose, K, a, bb)
        loss.backward(gradient=one)
        x = pn._side
        dn.join_side(); pn.join_side()
        x.wait_stream(main)
        with torch.cuda.stream(x):
            opt.step()

    def step(early):
        opt.zero_grad()
        d_t, d_r, d_l = dn.forward_pair_split(frames)
        pose, a, bb = pn(tgt, ref, d_t, d_r)
        loss = Fh.photometric_loss(tgt, ref, d_l, pose, K, a, bb)
        loss.backward(gradient=one)
        if early and pn._side is not None:
            x = pn._side
            with torch.cuda.stream(x):
                x.wait_event(ev)
                dn._join_pending = pn._join_pending = False      # (timing probe: the update does not wait for the late gradients)
                opt.step()
            main.wait_stream(x)
            main.wait_stream(dn._side)
        else:
            opt.step()

    def timed(early, n):
        evs = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
        evs[0].record()
        for i in range(n):
            step(early)
            evs[i + 1].record()
        torch.cuda.synchronize()
        t = sorted(evs[i].elapsed_time(evs[i + 1]) for i in range(n))
        return t[len(t) // 2]

    for _ in range(10):
        step(False)
    torch.cuda.synchronize()
    dn.grad_ready_hook = hook          # both modes run with the hook (the program is split at that layer either way)
    for _ in range(10):
        step(True)
    torch.cuda.synchronize()
    for r in range(3):
        print(f"round {r}: base {timed(False, args.steps):.4f} ms   early({args.layer}) {timed(True, args.steps):.4f} ms", flush=True)
    dn.grad_ready_hook = None
    global_step = step

    def timed_fn(fn, n):
        evs = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
        evs[0].record()
        for i in range(n):
            fn()
            evs[i + 1].record()
        torch.cuda.synchronize()
        t = sorted(evs[i].elapsed_time(evs[i + 1]) for i in range(n))
        return t[len(t) // 2]

    for _ in range(10):
        step_next()
    torch.cuda.synchronize()
    for r in range(3):
        print(f"round {r}: base {timed_fn(lambda: global_step(False), args.steps):.4f} ms   update under the next forward pass "
              f"{timed_fn(step_next, args.steps):.4f} ms", flush=True)


if __name__ == "__main__":
    main()
