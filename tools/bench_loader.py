#!/usr/bin/env python3
"""Throughput of the frame-pair input pipeline (coivo_amd.data.PairLoader) on synthetic sequence folders:
   python tools/bench_loader.py [npy|png] [workers=8] [native=256x320] [batch=8] [prefetch=3]
Writes a temporary dataset (under $TMPDIR), iterates two epochs and prints pairs/s of the second one -- with nothing else on
the GPU, so this is the rate the loader can deliver, to be read against the training step's rate.  With a sixth argument
`train` every batch is also trained on (bf16 DCDP+LCC step, random weights): the end-to-end rate of loader + step in one process.
DECODERS=n decodes images in n worker processes; OWN_COPY_STREAM=0 / 1 overrides the loader's choice of stream for the upload (see data.PairLoader)."""
import os
import shutil
import sys
import tempfile
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from coivo_amd.data import PairLoader, SequenceFolder  # noqa: E402


def main():
    kind = sys.argv[1] if len(sys.argv) > 1 else "npy"
    workers = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    h, w = (int(v) for v in (sys.argv[3] if len(sys.argv) > 3 else "256x320").split("x"))
    B = int(sys.argv[4]) if len(sys.argv) > 4 else 8
    prefetch = int(sys.argv[5]) if len(sys.argv) > 5 else 2
    train = len(sys.argv) > 6 and sys.argv[6] == "train"
    root = tempfile.mkdtemp(prefix="colvo_loader_")
    try:
        rng = np.random.default_rng(0)
        nseq, nfr = 4, 129                                   # 4 x 128 pairs
        for s in range(nseq):
            d = os.path.join(root, f"seq{s}")
            os.makedirs(d)
            for k in range(nfr):
                a = rng.integers(0, 256, size=(h, w, 3), dtype=np.uint8)
                if kind == "npy":
                    np.save(os.path.join(d, f"{k:05d}.npy"), a)
                else:
                    from PIL import Image
                    Image.fromarray(a).save(os.path.join(d, f"{k:05d}.png"), compress_level=1)
        ds = SequenceFolder(root)
        ld = PairLoader(ds, B, (256, 320), shuffle=True, workers=workers, prefetch=prefetch,
                        own_copy_stream={None: None, "0": False, "1": True}[os.environ.get("OWN_COPY_STREAM")],
                        decoders=int(os.environ.get("DECODERS", "0")))
        if os.environ.get("KEEP_AUX"):       # probe: keep the auxiliary side stream although the loader claimed a queue
            from coivo_amd import streams
            streams.configure(0)
        step = None
        if train:
            from coivo_amd import functional as Fh, nn as hnn
            from coivo_amd.optim import FusedAdam
            dev = torch.device("cuda")
            dn, pn = hnn.DepthNet(compute_dtype=torch.bfloat16, device=dev), hnn.PoseNet(compute_dtype=torch.bfloat16, device=dev)
            opt = FusedAdam([dn, pn], lr=1e-4)
            one = torch.ones((), device=dev)

            def step(b):
                opt.zero_grad()
                frames = b["frames"]                 # the loader's stacked buffer: [target frames | reference frames]
                d_t, d_r, d_l = dn.forward_pair_split(frames)
                pose, a, bb = pn(frames[:B], frames[B:], d_t, d_r)
                Fh.photometric_loss(frames[:B], frames[B:], d_l, pose, b["K"], a, bb).backward(gradient=one)
                opt.step()
        for ep in range(3 if train else 2):
            ld.set_epoch(ep)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            n = 0
            for batch in ld:
                n += batch["tgt"].shape[0]
                if step is not None:
                    step(batch)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            print(f"{kind} {h}x{w} batch {B} workers {workers} prefetch {prefetch}{' +train' if train else ''} epoch {ep}: {n} pairs in {dt * 1e3:.1f} ms = {n / dt:.0f} pairs/s", flush=True)
    finally:
        shutil.rmtree(root, ignore_errors=True)


if __name__ == "__main__":
    main()
