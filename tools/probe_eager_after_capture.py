#!/usr/bin/env python3
"""Why do eager steps get slow after a hipGraph capture in the same process (1.5 -> 4.6 ms)?

    python3 tools/probe_eager_after_capture.py [--pairs 8] [--steps 30]
    rocprofv3 --kernel-trace -d gpurun_out/eac -- python3 tools/probe_eager_after_capture.py --mark      (queues per phase)

Phases, each timed over --steps steps (wall clock between two device synchronisations):
  eager0            eager steps in a fresh process
  replay            the captured step replayed
  eager_after       eager steps again, graph object alive
  eager_graph_freed eager steps after `del` of the graph + empty_cache (does releasing the exec's streams help?)
  eager_new_side    eager steps after giving both networks FRESH weight-gradient side streams
With --mark every phase is bracketed by a tiny marker kernel launch count (torch.zeros fill of a distinct size) so that a kernel
trace can be cut into phases (tools/step_timeline.py prints the HSA queue of every kernel).
"""
import argparse
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--pairs", type=int, default=8)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--mark", action="store_true")
    ap.add_argument("--carry", type=int, default=1)
    ap.add_argument("--policy", type=int, default=2)
    ap.add_argument("--group", type=int, default=2)
    ap.add_argument("--capture-first", action="store_true", help="capture before ANY eager step has run in the process (bench.py --graph on)")
    ap.add_argument("--only-eager", action="store_true", help="time eager steps in a fresh process and stop")
    args = ap.parse_args()
    from coivo_amd import build
    build.ensure()
    from coivo_amd import functional as Fh, nn as hnn, synth
    from coivo_amd.graph import GraphedTrainStep
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
    opt = FusedAdam([dn, pn], lr=1e-4)
    b = synth.make_batch(B, H, W, seed=1234, device=dev)
    frames = torch.cat([b["tgt"], b["ref"]])
    tgt, ref, K = frames[:B], frames[B:], b["K"]
    one = torch.ones((), device=dev)

    def eager():
        opt.zero_grad()
        d_t, d_r, d_l = dn.forward_pair_split(frames)
        pose, a, bb = pn(tgt, ref, d_t, d_r)
        loss = Fh.photometric_loss(tgt, ref, d_l, pose, K, a, bb)
        loss.backward(gradient=one)
        opt.step()

    marks = [0]

    def timed(tag, fn):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        if args.mark:
            marks[0] += 1
            torch.zeros(1000 + marks[0], device=dev)       # marker: a fill kernel of a distinct size
            torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(args.steps):
            fn()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t) / args.steps * 1e3
        print(f"{tag:20s} {ms:8.3f} ms/step", flush=True)
        return ms

    r = {}
    if not args.capture_first:
        r["eager0"] = timed("eager0", eager)
    if args.only_eager:
        import json
        print("PROBE_JSON " + json.dumps({k: round(v, 4) for k, v in r.items()}))
        return
    step = GraphedTrainStep(dn, pn, opt, B, H, W, carry=bool(args.carry), capture_policy=args.policy, capture_group=args.group)
    step.frames.copy_(frames)
    step.K.copy_(K)
    step.capture()
    print("graph stats:", step.stats, flush=True)
    r["replay"] = timed("replay", step)
    r["eager_after"] = timed("eager_after", eager)
    step.graph = None
    del step
    import gc
    gc.collect()
    torch.cuda.empty_cache()
    torch.cuda.synchronize()
    r["eager_graph_freed"] = timed("eager_graph_freed", eager)
    for net in (dn, pn):
        net._side = None
    r["eager_new_side"] = timed("eager_new_side", eager)
    import json
    print("PROBE_JSON " + json.dumps({k: round(v, 4) for k, v in r.items()}))


if __name__ == "__main__":
    main()
