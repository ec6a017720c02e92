"""Generate tests/golden/*.npz from the oracle (run here, committed with its outputs).

There are no upstream vectors (parity unpinned, SPEC.md); these fixtures pin the
oracle itself against drift and give the GPU box fixed inputs -> expected outputs.

    python oracle/make_golden.py
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from coivo_amd import synth  # noqa: E402
from oracle import colvo_spec as S  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")


def loss_case(name, B, H, W, seed, jitter=True):
    """Inputs + loss + gradients of the fused path (a3..a7) at the synthetic GT-ish operating point."""
    b = synth.make_batch(B, H, W, seed=seed)
    g = torch.Generator().manual_seed(seed + 7)
    depth = (b["gt_depth"] * (1 + 0.05 * torch.randn(B, 1, H, W, generator=g))).clamp(0.2, 9.0) if jitter else b["gt_depth"]
    pose = b["gt_pose"] + (0.005 * torch.randn(B, 6, generator=g) if jitter else 0)
    a, bb = b["gt_a"].clone(), b["gt_b"].clone()
    leaves = [t.clone().requires_grad_(True) for t in (depth, pose, a, bb)]
    loss = S.photometric_loss(b["tgt"], b["ref"], leaves[0], leaves[1], b["K"], leaves[2], leaves[3])
    grads = torch.autograd.grad(loss, leaves)
    warped, valid = S.inverse_warp(b["ref"], depth, pose, b["K"])
    np.savez_compressed(
        os.path.join(OUT, name + ".npz"),
        tgt=b["tgt"].numpy(), ref=b["ref"].numpy(), K=b["K"].numpy(),
        depth=depth.numpy(), pose=pose.numpy(), lcc_a=a.numpy(), lcc_b=bb.numpy(),
        loss=np.float32(loss.item()), valid_frac=np.float32(valid.mean().item()),
        warped_sum=np.float64(warped.double().sum().item()),
        d_depth=grads[0].numpy(), d_pose=grads[1].numpy(), d_a=grads[2].numpy(), d_b=grads[3].numpy())
    print(name, "loss", loss.item(), "valid", valid.mean().item())


GRAD_SAMPLE = 4096


def grad_digest(g: torch.Tensor):
    """What the fixtures keep of one parameter gradient: a fixed strided sample of <= GRAD_SAMPLE elements (the whole
    tensor when it is that small) and [sum, L2 norm, max |.|] over ALL elements in float64."""
    f = g.detach().flatten()
    stride = max(1, -(-f.numel() // GRAD_SAMPLE))
    d = f.double()
    return f[::stride][:GRAD_SAMPLE].clone(), torch.stack([d.sum(), d.norm(), d.abs().max()])


def net_case(name, B, H, W, seed):
    """Seeds + expected network outputs (weights are regenerated from the seed, not stored) + a digest of EVERY
    parameter gradient of the coupled step, from the fp32 oracle (gs32_<net>.<param>: strided sample) and from the same
    step evaluated in fp64 (gs64_: sample, gn64_: sum / norm / max) -- the fp64 values are the yardstick the GPU test
    measures both the HIP path and the fp32 oracle against (SPEC.md §7)."""
    b = synth.make_batch(B, H, W, seed=seed)
    dn, pn = S.make_models(seed=seed)
    loss, d_t, d_r, pose, a, bb = S.dcdp_forward(dn, pn, b["tgt"], b["ref"], b["K"])
    loss.backward()
    dn64, pn64 = S.make_models(seed=seed, dtype=torch.float64)
    loss64 = S.dcdp_forward(dn64, pn64, b["tgt"].double(), b["ref"].double(), b["K"].double())[0]
    loss64.backward()
    digests = {}
    for tag, net, net64 in (("depth", dn, dn64), ("pose", pn, pn64)):
        for (pname, p), (_, p64) in zip(net.named_parameters(), net64.named_parameters()):
            digests[f"gs32_{tag}.{pname}"] = grad_digest(p.grad)[0].numpy()
            smp, nrm = grad_digest(p64.grad)
            digests[f"gs64_{tag}.{pname}"] = smp.numpy()
            digests[f"gn64_{tag}.{pname}"] = nrm.numpy()
    np.savez_compressed(
        os.path.join(OUT, name + ".npz"),
        B=B, H=H, W=W, seed=seed,
        loss=np.float32(loss.item()), loss64=np.float64(loss64.item()), depth_t=d_t.detach().numpy(),
        depth_r=d_r.detach().numpy(), pose=pose.detach().numpy(), lcc_a=a.detach().numpy(), lcc_b=bb.detach().numpy(),
        **digests)
    print(name, "loss", loss.item(), "fp64", loss64.item(), "gradient digests", len(digests) // 3)


def terms_case(name, B, H, W, seed):
    """Inputs + values + gradients of the widened-objective terms (SPEC.md §6b): geometric consistency, smoothness,
    multi-scale photometric."""
    b = synth.make_batch(B, H, W, seed=seed)
    g = torch.Generator().manual_seed(seed + 7)
    d_t = (b["gt_depth"] * (1 + 0.05 * torch.randn(B, 1, H, W, generator=g))).clamp(0.2, 9.0)
    d_r = (b["gt_depth"] * (1 + 0.05 * torch.randn(B, 1, H, W, generator=g))).clamp(0.2, 9.0)
    pose = b["gt_pose"] + 0.005 * torch.randn(B, 6, generator=g)
    a, bb = b["gt_a"].clone(), b["gt_b"].clone()
    lt, lr, lp, la, lb = (t.clone().requires_grad_(True) for t in (d_t, d_r, pose, a, bb))
    geo = S.geometric_consistency_loss(lt, lr, lp, b["K"])
    ggeo = torch.autograd.grad(geo, (lt, lr, lp))
    sm = S.smoothness_loss(lt, b["tgt"])
    gsm = torch.autograd.grad(sm, lt)[0]
    ms = S.multiscale_photometric_loss(b["tgt"], b["ref"], lt, lp, b["K"], la, lb)
    gms = torch.autograd.grad(ms, (lt, lp, la, lb))
    full = S.dcdp_full_loss(b["tgt"], b["ref"], lt, lr, lp, b["K"], la, lb)
    gfull = torch.autograd.grad(full, (lt, lr, lp, la, lb))
    np.savez_compressed(
        os.path.join(OUT, name + ".npz"),
        tgt=b["tgt"].numpy(), ref=b["ref"].numpy(), K=b["K"].numpy(), depth_t=d_t.numpy(), depth_r=d_r.numpy(),
        pose=pose.numpy(), lcc_a=a.numpy(), lcc_b=bb.numpy(),
        geo=np.float32(geo.item()), geo_d_t=ggeo[0].numpy(), geo_d_r=ggeo[1].numpy(), geo_d_pose=ggeo[2].numpy(),
        smooth=np.float32(sm.item()), smooth_d_t=gsm.numpy(),
        ms=np.float32(ms.item()), ms_d_t=gms[0].numpy(), ms_d_pose=gms[1].numpy(), ms_d_a=gms[2].numpy(), ms_d_b=gms[3].numpy(),
        full=np.float32(full.item()), full_d_t=gfull[0].numpy(), full_d_r=gfull[1].numpy(), full_d_pose=gfull[2].numpy(),
        full_d_a=gfull[3].numpy(), full_d_b=gfull[4].numpy())
    print(name, "geo", geo.item(), "smooth", sm.item(), "ms", ms.item(), "full", full.item())


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    torch.set_num_threads(1)  # one thread: bit-stable reductions
    loss_case("loss_b2_32x40", 2, 32, 40, seed=11)
    loss_case("loss_b2_64x96", 2, 64, 96, seed=12)
    loss_case("loss_b1_256x320", 1, 256, 320, seed=13)
    loss_case("loss_b2_33x47_ragged", 2, 33, 47, seed=14)   # not tile-aligned
    terms_case("terms_b2_48x64", 2, 48, 64, seed=15)
    net_case("net_b2_64x96", 2, 64, 96, seed=21)
    net_case("net_b1_32x64", 1, 32, 64, seed=22)
