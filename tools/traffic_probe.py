#!/usr/bin/env python3
"""Workload for the HBM-traffic PMC passes: the fused loss fwd+bwd at configs[2] and configs[1] shapes, plus
a streaming kernel of exactly known byte count (k_adam: 16 B read + 12 B written per element, dword per lane)
in the same process, which calibrates FETCH_SIZE / WRITE_SIZE for this access width on gfx950."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from coivo_amd import functional as Fh, ops, synth  # noqa: E402

dev = torch.device("cuda:0")
N = 64 * 1024 * 1024      # 256 MiB per array: past the 256 MiB Infinity Cache in aggregate (4 arrays = 1 GiB)
p, g, m, v = (torch.randn(N, device=dev) for _ in range(4))
v.abs_()
step = torch.zeros(1, dtype=torch.int32, device=dev)
for _ in range(3):
    ops.adam_step(p, g, m, v, step, lr=1e-4, beta1=0.9, beta2=0.999, eps=1e-8)
torch.cuda.synchronize()
del p, g, m, v
for (B, H, W) in ((32, 512, 640), (8, 256, 320)):
    b = synth.make_batch(min(B, 4), H, W, seed=77, device=dev)
    rep = lambda t: t.repeat(B // min(B, 4), *([1] * (t.dim() - 1))).contiguous()
    tgt, ref, K = rep(b["tgt"]), rep(b["ref"]), rep(b["K"])
    leaves = [rep(b[k]).requires_grad_(True) for k in ("gt_depth", "gt_pose", "gt_a", "gt_b")]
    # the training form of the op, as in bench.py roofline_cfg2: gradients leave unnormalised with two device scalars
    # (functional.GradHandover), so the backward call launches nothing
    hand, hand_p = Fh.GradHandover(), Fh.GradHandover()
    leaves[0]._colvo_handover = hand
    leaves[1]._colvo_handover = leaves[2]._colvo_handover = leaves[3]._colvo_handover = hand_p
    for _ in range(5):
        loss = Fh.photometric_loss(tgt, ref, leaves[0], leaves[1], K, leaves[2], leaves[3])
        g = torch.autograd.grad(loss, leaves)
        hand.take((g[0],))
        hand_p.take(tuple(g[1:]))
    torch.cuda.synchronize()
print("probe done")
