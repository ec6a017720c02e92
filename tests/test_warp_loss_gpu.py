"""-m gpu parity of the fused project->sample->LCC->SSIM/L1 kernels (a3..a7) against the oracle.

Tolerances are BASELINE.json's: 1e-5 (abs) on the scalar loss in fp32.
"""
import os

import pytest
import torch

from coivo_amd import synth
from tests.gpu_util import assert_close_frac, dev, load_npz, to_dev

pytestmark = pytest.mark.gpu

LOSS_TOL = 1e-5


def _hip_loss_and_grads(t):
    from coivo_amd import functional as Fh
    leaves = [t[k].clone().requires_grad_(True) for k in ("depth", "pose", "lcc_a", "lcc_b")]
    loss = Fh.photometric_loss(t["tgt"], t["ref"], leaves[0], leaves[1], t["K"], leaves[2], leaves[3])
    grads = torch.autograd.grad(loss, leaves)
    return loss, grads


def _oracle_loss_and_grads(t):
    from oracle import colvo_spec as S
    leaves = [t[k].clone().requires_grad_(True) for k in ("depth", "pose", "lcc_a", "lcc_b")]
    loss = S.photometric_loss(t["tgt"], t["ref"], leaves[0], leaves[1], t["K"], leaves[2], leaves[3])
    grads = torch.autograd.grad(loss, leaves)
    return loss, grads


def _compare(hl, hg, ol, og, tag):
    assert abs(hl.item() - ol.item()) < LOSS_TOL, (tag, hl.item(), ol.item())
    assert_close_frac(hg[0], og[0], rtol=2e-3, atol_scale=2e-4, max_bad_frac=2e-4, what=tag + " d_depth")
    # d_pose / d_a / d_b are sums over all pixels of signed terms (heavy cancellation): the fp32 oracle itself
    # sits ~4e-4 (of the largest component) away from its fp64 evaluation, so 3e-3 is the meaningful bar
    assert_close_frac(hg[1], og[1], rtol=2e-3, atol_scale=3e-3, max_bad_frac=0, what=tag + " d_pose")
    assert_close_frac(hg[2], og[2], rtol=2e-3, atol_scale=3e-3, max_bad_frac=0, what=tag + " d_a")
    assert_close_frac(hg[3], og[3], rtol=2e-3, atol_scale=3e-3, max_bad_frac=0, what=tag + " d_b")


@pytest.mark.parametrize("name", ["loss_b2_32x40", "loss_b2_64x96", "loss_b1_256x320", "loss_b2_33x47_ragged"])
def test_golden_fixture(golden_dir, name):
    g = load_npz(os.path.join(golden_dir, name + ".npz"))
    t = to_dev(g, ("tgt", "ref", "K", "depth", "pose", "lcc_a", "lcc_b"))
    hl, hg = _hip_loss_and_grads(t)
    _compare(hl, hg, g["loss"], (g["d_depth"], g["d_pose"], g["d_a"], g["d_b"]), name)


@pytest.mark.parametrize("name", ["loss_b2_64x96", "loss_b2_33x47_ragged"])
def test_golden_fixture_two_pass(golden_dir, name, monkeypatch):
    """The un-fused form (forward kernel, then the backward kernel that re-evaluates the warp) stays covered."""
    from coivo_amd import functional as Fh
    monkeypatch.setattr(Fh, "FUSE_TRAINING_PASS", False)
    g = load_npz(os.path.join(golden_dir, name + ".npz"))
    t = to_dev(g, ("tgt", "ref", "K", "depth", "pose", "lcc_a", "lcc_b"))
    hl, hg = _hip_loss_and_grads(t)
    _compare(hl, hg, g["loss"], (g["d_depth"], g["d_pose"], g["d_a"], g["d_b"]), name + " two-pass")


def test_fused_pass_equals_two_pass_and_scales_with_grad_loss(monkeypatch):
    """One-pass loss+gradient == forward then backward (same partial-sum orders are not guaranteed: tolerance), and the
    incoming dL/dloss is applied (2.5 x loss -> 2.5 x gradients)."""
    from coivo_amd import functional as Fh
    monkeypatch.setattr(Fh, "FUSE_TRAINING_PASS", True)      # whatever COLVO_LOSS_UNFUSED says
    t = to_dev(_case(2, 96, 128, 120, 1.0))

    def run(scale):
        leaves = [t[k].clone().requires_grad_(True) for k in ("depth", "pose", "lcc_a", "lcc_b")]
        loss = Fh.photometric_loss(t["tgt"], t["ref"], leaves[0], leaves[1], t["K"], leaves[2], leaves[3])
        return loss.detach(), torch.autograd.grad(loss * scale, leaves)

    lf, gf = run(1.0)
    _, gf25 = run(2.5)
    monkeypatch.setattr(Fh, "FUSE_TRAINING_PASS", False)
    lu, gu = run(1.0)
    assert abs(lf.item() - lu.item()) < 1e-6
    for a, b, c, nm in zip(gf, gu, gf25, ("d_depth", "d_pose", "d_a", "d_b")):
        # the reduced gradients are cancelling sums: scaling every term (two-pass) or the total (one-pass) moves them by
        # ~1e-4 of their own size, far inside the 3e-3 bar against the oracle
        rt, at = (1e-4, 1e-5) if nm == "d_depth" else (1e-3, 1e-3)
        assert_close_frac(a, b, rtol=rt, atol_scale=at, max_bad_frac=0, what="fused vs two-pass " + nm)
        assert_close_frac(c, 2.5 * a, rtol=1e-5, atol_scale=1e-6, max_bad_frac=0, what="grad_loss scaling " + nm)


def _case(B, H, W, seed, pose_scale=1.0):
    b = synth.make_batch(B, H, W, seed=seed)
    g = torch.Generator().manual_seed(seed + 1)
    t = dict(tgt=b["tgt"], ref=b["ref"], K=b["K"],
             depth=(b["gt_depth"] * (1 + 0.05 * torch.randn(B, 1, H, W, generator=g))).clamp(0.2, 9.0),
             pose=b["gt_pose"] * pose_scale + 0.003 * torch.randn(B, 6, generator=g),
             lcc_a=b["gt_a"], lcc_b=b["gt_b"])
    return t


@pytest.mark.parametrize("B,H,W,seed,ps", [
    (2, 256, 320, 101, 1.0),     # BASELINE configs[0] shape
    (3, 64, 64, 102, 4.0),       # large motion: many out-of-view pixels
    (1, 17, 65, 103, 1.0),       # one pixel past a tile edge in both directions
    (2, 16, 64, 104, 1.0),       # exactly one tile
    (1, 2, 2, 105, 0.0),         # smallest legal image (reflect pad needs >= 2)
    (2, 50, 130, 106, 2.0),
])
def test_live_oracle(B, H, W, seed, ps):
    t = _case(B, H, W, seed, ps)
    ol, og = _oracle_loss_and_grads(t)
    hl, hg = _hip_loss_and_grads(to_dev(t))
    _compare(hl, hg, ol, og, f"B{B} {H}x{W}")


def test_all_invalid_is_zero_loss_and_zero_grads():
    t = _case(1, 32, 64, 107)
    t["pose"] = torch.tensor([[0.0, 0.0, -100.0, 0.0, 0.0, 0.0]])   # everything behind the camera
    hl, hg = _hip_loss_and_grads(to_dev(t))
    assert hl.item() == 0.0
    for g in hg:
        assert torch.count_nonzero(g) == 0


def test_inverse_warp_matches_oracle():
    from coivo_amd import functional as Fh
    from oracle import colvo_spec as S
    t = _case(2, 48, 80, 108, 3.0)
    w_o, v_o = S.inverse_warp(t["ref"], t["depth"], t["pose"], t["K"])
    d = to_dev(t)
    w_h, v_h = Fh.inverse_warp(d["ref"], d["depth"], d["pose"], d["K"])
    assert (v_h.cpu() != v_o).float().mean().item() <= 2e-4
    assert_close_frac(w_h, w_o, rtol=1e-5, atol_scale=1e-5, max_bad_frac=2e-4, what="warped")
    assert 0.3 < v_o.mean().item() < 1.0


def test_full_size_properties():
    """BASELINE configs[2] size (B=32, 640x512): size-independent properties instead of the oracle."""
    from coivo_amd import functional as Fh
    B, H, W = 32, 512, 640
    b = to_dev(synth.make_batch(B, H, W, seed=109))
    ident = torch.zeros(B, 6, device=dev())
    one, zero = torch.ones(B, 1, device=dev()), torch.zeros(B, 1, device=dev())
    # (1) identical frames + identity pose -> ~0 (border pixels may flip invalid, see test_oracle.py)
    l0 = Fh.photometric_loss(b["tgt"], b["tgt"], b["gt_depth"], ident, b["K"], one, zero)
    assert 0.0 <= l0.item() < 2e-3
    # (2) ground-truth parameters beat identity parameters
    lg = Fh.photometric_loss(b["tgt"], b["ref"], b["gt_depth"], b["gt_pose"], b["K"], b["gt_a"], b["gt_b"])
    li = Fh.photometric_loss(b["tgt"], b["ref"], b["gt_depth"], ident, b["K"], one, zero)
    assert lg.item() < li.item()
    # (3) bitwise determinism + backward is linear in grad_loss
    leaves = [b[k].clone().requires_grad_(True) for k in ("gt_depth", "gt_pose", "gt_a", "gt_b")]

    def run(scale):
        loss = Fh.photometric_loss(b["tgt"], b["ref"], leaves[0], leaves[1], b["K"], leaves[2], leaves[3])
        return loss, torch.autograd.grad(loss * scale, leaves)

    l1, g1 = run(1.0)
    l2, g2 = run(1.0)
    l3, g3 = run(2.0)
    assert torch.equal(l1, l2)
    for a, c in zip(g1, g2):
        assert torch.equal(a, c)
    for a, c in zip(g1, g3):
        assert torch.allclose(2 * a, c, rtol=1e-6, atol=0)
    assert all(torch.isfinite(g).all() for g in g1)
    # (4) batch-split consistency: loss over the batch = valid-weighted mean of per-half losses
    h = B // 2
    parts = []
    for sl in (slice(0, h), slice(h, B)):
        from coivo_amd import _lib
        loss = Fh.photometric_loss(b["tgt"][sl], b["ref"][sl], b["gt_depth"][sl], b["gt_pose"][sl], b["K"][sl],
                                   b["gt_a"][sl], b["gt_b"][sl])
        _, v = Fh.inverse_warp(b["ref"][sl], b["gt_depth"][sl], b["gt_pose"][sl], b["K"][sl])
        parts.append((loss.item(), v.sum().item()))
    comb = (parts[0][0] * parts[0][1] + parts[1][0] * parts[1][1]) / (parts[0][1] + parts[1][1])
    assert abs(comb - lg.item()) < 1e-6


def test_rejects_cpu_tensors_and_bad_shapes():
    from coivo_amd import functional as Fh
    t = _case(1, 16, 16, 110)
    with pytest.raises(RuntimeError):
        Fh.photometric_loss(t["tgt"], t["ref"], t["depth"], t["pose"], t["K"], t["lcc_a"], t["lcc_b"])
    d = to_dev(t)
    with pytest.raises(ValueError):
        Fh.photometric_loss(d["tgt"], d["ref"], d["depth"][:, :, :8], d["pose"], d["K"], d["lcc_a"], d["lcc_b"])


def test_deferred_normalisation_handover():
    """The training form (DepthNet.forward_pair_split): photometric_loss returns the depth gradient UNNORMALISED and posts
    dL/dloss and 1/max(3 n_valid, 1) as device scalars; raw * scales must be the oracle's gradient, pose / LCC gradients
    arrive normalised, and a gradient that does not come straight from the loss is refused."""
    from coivo_amd import functional as Fh
    t = to_dev(_case(2, 96, 128, 121, 1.0))
    ol, og = _oracle_loss_and_grads({k: v.cpu() for k, v in t.items()})
    leaves = [t[k].clone().requires_grad_(True) for k in ("depth", "pose", "lcc_a", "lcc_b")]
    hand = Fh.GradHandover()
    leaves[0]._colvo_handover = hand
    loss = Fh.photometric_loss(t["tgt"], t["ref"], leaves[0], leaves[1], t["K"], leaves[2], leaves[3])
    up = torch.full((), 2.5, device=loss.device)
    graw, gp, ga, gb = torch.autograd.grad(loss, leaves, grad_outputs=up)
    sa, sb = hand.take((graw,))
    assert sa is not None and sb is not None and hand.take((None,)) == (None, None)
    assert abs(sa.item() - 2.5) < 1e-7
    d_depth = graw * (sa * sb)
    _compare(loss, (d_depth / 2.5, gp / 2.5, ga / 2.5, gb / 2.5), ol, og, "deferred")
    # the mailbox refuses a gradient that is not the tensor the loss handed over
    loss = Fh.photometric_loss(t["tgt"], t["ref"], leaves[0], leaves[1], t["K"], leaves[2], leaves[3])
    graw = torch.autograd.grad(loss, leaves)[0]
    with pytest.raises(RuntimeError):
        hand.take((graw + 0.0,))


def test_handover_refuses_second_use_created_before_the_loss():
    """d_l (forward_pair_split's third output) used in a second term that was created BEFORE the loss.  The loss node
    runs first in backward, so its raw gradient is the first arrival in the producer's input buffer -- where autograd would add
    the second gradient IN PLACE (same address) if the mailbox did not hold the raw tensor.  Must raise, never scale (raw + other)."""
    from coivo_amd import functional as Fh
    from coivo_amd import nn as hnn
    from coivo_amd import synth
    dn, pn = hnn.DepthNet(compute_dtype=torch.float32), hnn.PoseNet(compute_dtype=torch.float32)
    d = to_dev(synth.make_batch(1, 32, 64, seed=5))
    frames = torch.cat([d["tgt"], d["ref"]])
    d_t, d_r, d_l = dn.forward_pair_split(frames)
    extra = (d_l * d_l).mean()
    pose, a, b = pn(d["tgt"], d["ref"], d_t, d_r)
    loss = Fh.photometric_loss(d["tgt"], d["ref"], d_l, pose, d["K"], a, b)
    with pytest.raises(RuntimeError, match="reserved for photometric_loss"):
        (loss + extra).backward()
    torch.cuda.synchronize()
    # the networks stay usable: the ordinary route (a plain tensor for the second use) gives finite gradients
    dn.zero_grad(); pn.zero_grad()
    d_t, d_r, d_l = dn.forward_pair_split(frames)
    pose, a, b = pn(d["tgt"], d["ref"], d_t, d_r)
    loss = Fh.photometric_loss(d["tgt"], d["ref"], d_l, pose, d["K"], a, b) + (d_t * d_t).mean()
    loss.backward()
    torch.cuda.synchronize()
    assert torch.isfinite(dn.flat_grad).all() and dn.flat_grad.abs().max() > 0
