"""-m gpu parity of the widened-objective terms against the oracle: geometric consistency,
edge-aware smoothness, 2x2 pooling / multi-scale photometric, and the whole dcdp_full_loss step."""
import os

import pytest
import torch

from coivo_amd import synth
from tests.gpu_util import assert_close_frac, dev, load_npz, to_dev

pytestmark = pytest.mark.gpu


def _terms(golden_dir):
    return load_npz(os.path.join(golden_dir, "terms_b2_48x64.npz"))


def test_geometric_consistency_golden(golden_dir):
    from coivo_amd import functional as Fh
    g = _terms(golden_dir)
    t = to_dev(g, ("depth_t", "depth_r", "pose", "K"))
    dt, dr, pose = (t[k].clone().requires_grad_(True) for k in ("depth_t", "depth_r", "pose"))
    loss = Fh.geometric_consistency_loss(dt, dr, pose, t["K"])
    assert abs(loss.item() - g["geo"].item()) < 1e-5
    gt, gr, gp = torch.autograd.grad(loss * 2.0, (dt, dr, pose))
    assert_close_frac(gt / 2, g["geo_d_t"], rtol=2e-3, atol_scale=2e-4, max_bad_frac=2e-4, what="geo d_depth_t")
    assert_close_frac(gr / 2, g["geo_d_r"], rtol=2e-3, atol_scale=2e-4, max_bad_frac=2e-4, what="geo d_depth_r")
    assert_close_frac(gp / 2, g["geo_d_pose"], rtol=2e-3, atol_scale=3e-3, max_bad_frac=0, what="geo d_pose")


@pytest.mark.parametrize("B,H,W,seed", [(1, 2, 2, 61), (2, 33, 47, 62), (4, 256, 320, 63)])
def test_geometric_consistency_live_oracle(B, H, W, seed):
    from coivo_amd import functional as Fh
    from oracle import colvo_spec as S
    b = synth.make_batch(B, H, W, seed=seed)
    g = torch.Generator().manual_seed(seed)
    d_t = (b["gt_depth"] * (1 + 0.05 * torch.randn(B, 1, H, W, generator=g))).clamp(0.2, 9.0)
    d_r = (b["gt_depth"] * (1 + 0.05 * torch.randn(B, 1, H, W, generator=g))).clamp(0.2, 9.0)
    lo = [t.clone().requires_grad_(True) for t in (d_t, d_r, b["gt_pose"])]
    lh = [t.clone().to(dev()).requires_grad_(True) for t in (d_t, d_r, b["gt_pose"])]
    ol = S.geometric_consistency_loss(*lo, b["K"])
    hl = Fh.geometric_consistency_loss(*lh, b["K"].to(dev()))
    assert abs(hl.item() - ol.item()) < 1e-5
    og, hg = torch.autograd.grad(ol, lo), torch.autograd.grad(hl, lh)
    assert_close_frac(hg[0], og[0], rtol=2e-3, atol_scale=2e-4, max_bad_frac=5e-4, what="geo d_depth_t")
    assert_close_frac(hg[1], og[1], rtol=2e-3, atol_scale=2e-4, max_bad_frac=5e-4, what="geo d_depth_r")
    assert_close_frac(hg[2], og[2], rtol=2e-3, atol_scale=3e-3, max_bad_frac=0, what="geo d_pose")


def test_smoothness_golden_and_ragged(golden_dir):
    from coivo_amd import functional as Fh
    from oracle import colvo_spec as S
    g = _terms(golden_dir)
    t = to_dev(g, ("depth_t", "tgt"))
    dt = t["depth_t"].clone().requires_grad_(True)
    loss = Fh.smoothness_loss(dt, t["tgt"])
    assert abs(loss.item() - g["smooth"].item()) < 1e-6
    gd = torch.autograd.grad(loss, dt)[0]
    assert_close_frac(gd, g["smooth_d_t"], rtol=1e-3, atol_scale=1e-4, max_bad_frac=1e-4, what="smooth d_depth")
    b = synth.make_batch(3, 17, 23, seed=64)
    do = b["gt_depth"].clone().requires_grad_(True)
    dh = b["gt_depth"].clone().to(dev()).requires_grad_(True)
    lo, lh = S.smoothness_loss(do, b["tgt"]), Fh.smoothness_loss(dh, b["tgt"].to(dev()))
    assert abs(lo.item() - lh.item()) < 1e-6
    assert_close_frac(torch.autograd.grad(lh, dh)[0], torch.autograd.grad(lo, do)[0], rtol=1e-3, atol_scale=1e-4,
                      max_bad_frac=1e-4, what="smooth d_depth ragged")


def test_downsample2_and_multiscale(golden_dir):
    from coivo_amd import functional as Fh
    from oracle import colvo_spec as S
    g = _terms(golden_dir)
    x = torch.rand(2, 3, 12, 20)
    xh = x.to(dev()).requires_grad_(True)
    y = Fh.downsample2(xh)
    assert torch.allclose(y.cpu(), S.downsample2(x), atol=1e-7)
    gy = torch.rand_like(y)
    gx = torch.autograd.grad(y, xh, gy)[0]
    xo = x.clone().requires_grad_(True)
    assert torch.allclose(gx.cpu(), torch.autograd.grad(S.downsample2(xo), xo, gy.cpu())[0], atol=1e-7)
    with pytest.raises(ValueError):
        Fh.downsample2(torch.zeros(1, 1, 5, 6, device=dev()))
    t = to_dev(g, ("tgt", "ref", "K", "depth_t", "pose", "lcc_a", "lcc_b"))
    leaves = [t[k].clone().requires_grad_(True) for k in ("depth_t", "pose", "lcc_a", "lcc_b")]
    ms = Fh.multiscale_photometric_loss(t["tgt"], t["ref"], leaves[0], leaves[1], t["K"], leaves[2], leaves[3])
    assert abs(ms.item() - g["ms"].item()) < 1e-5
    hg = torch.autograd.grad(ms, leaves)
    assert_close_frac(hg[0], g["ms_d_t"], rtol=2e-3, atol_scale=3e-4, max_bad_frac=5e-4, what="ms d_depth")
    assert_close_frac(hg[1], g["ms_d_pose"], rtol=2e-3, atol_scale=3e-3, max_bad_frac=0, what="ms d_pose")
    assert_close_frac(hg[2], g["ms_d_a"], rtol=2e-3, atol_scale=3e-3, max_bad_frac=0, what="ms d_a")
    assert_close_frac(hg[3], g["ms_d_b"], rtol=2e-3, atol_scale=3e-3, max_bad_frac=0, what="ms d_b")


def test_full_loss_golden_and_step(golden_dir):
    """dcdp_full_loss on the fixture, then the whole widened step through the networks against the oracle."""
    from coivo_amd import functional as Fh
    from coivo_amd import nn as hnn
    from oracle import colvo_spec as S
    g = _terms(golden_dir)
    t = to_dev(g, ("tgt", "ref", "K", "depth_t", "depth_r", "pose", "lcc_a", "lcc_b"))
    leaves = [t[k].clone().requires_grad_(True) for k in ("depth_t", "depth_r", "pose", "lcc_a", "lcc_b")]
    full = Fh.dcdp_full_loss(t["tgt"], t["ref"], leaves[0], leaves[1], leaves[2], t["K"], leaves[3], leaves[4])
    assert abs(full.item() - g["full"].item()) < 1e-5
    hg = torch.autograd.grad(full, leaves)
    assert_close_frac(hg[0], g["full_d_t"], rtol=2e-3, atol_scale=3e-4, max_bad_frac=5e-4, what="full d_depth_t")
    assert_close_frac(hg[1], g["full_d_r"], rtol=2e-3, atol_scale=3e-4, max_bad_frac=5e-4, what="full d_depth_r")
    assert_close_frac(hg[2], g["full_d_pose"], rtol=2e-3, atol_scale=3e-3, max_bad_frac=0, what="full d_pose")
    B, H, W, seed = 2, 64, 96, 71
    dn_o, pn_o = S.make_models(seed)
    dn, pn = hnn.DepthNet(), hnn.PoseNet()
    dn.load_state_dict(dn_o.state_dict()); pn.load_state_dict(pn_o.state_dict())
    b = synth.make_batch(B, H, W, seed=seed)
    d = to_dev(b)
    lo = S.dcdp_forward(dn_o, pn_o, b["tgt"], b["ref"], b["K"], full_loss=True)[0]
    lh = hnn.dcdp_forward(dn, pn, d["tgt"], d["ref"], d["K"], full_loss=True)[0]
    assert abs(lo.item() - lh.item()) < 1e-5
    lo.backward(); lh.backward()
    for name in ("enc1a.weight", "iconv2.weight", "head.weight"):
        go, gh = dict(dn_o.named_parameters())[name].grad, dict(dn.named_parameters())[name].grad
        assert_close_frac(gh, go, rtol=1e-2, atol_scale=5e-3, max_bad_frac=1e-3, what=name)
    assert_close_frac(pn.pred.weight.grad, pn_o.pred.weight.grad, rtol=1e-2, atol_scale=5e-3, max_bad_frac=0, what="pose pred")


@pytest.mark.parametrize("B,H,W,geo,smooth,scales", [(2, 48, 64, 0.5, 0.1, 3), (1, 32, 32, 0.0, 0.1, 2), (3, 64, 96, 0.5, 0.0, 1),
                                                     (2, 40, 56, 0.0, 0.0, 1), (2, 40, 56, 0.0, 0.0, 3), (8, 256, 320, 0.5, 0.1, 3)])
def test_full_loss_one_node_equals_composite(B, H, W, geo, smooth, scales):
    """The one-node form of the widened objective against the term-by-term form: same value, same gradients."""
    from coivo_amd import functional as Fh
    b = synth.make_batch(B, H, W, seed=90 + B)
    d = to_dev(b)
    args = dict(geo_weight=geo, smooth_weight=smooth, num_scales=scales)

    def run(fn):
        leaves = [d[k].clone().requires_grad_(True) for k in ("gt_depth", "gt_pose", "gt_a", "gt_b")]
        d_r = (d["gt_depth"] * 1.07 + 0.05).clone().requires_grad_(True)
        loss = fn(d["tgt"], d["ref"], leaves[0], d_r, leaves[1], d["K"], leaves[2], leaves[3], **args)
        grads = torch.autograd.grad(loss * 3.0, leaves + [d_r], allow_unused=True)
        return loss, grads

    l1, g1 = run(Fh.dcdp_full_loss)
    l2, g2 = run(Fh.dcdp_full_loss_composite)
    assert abs(l1.item() - l2.item()) < 2e-6
    for a, c, name in zip(g1, g2, ("d_t", "pose", "a", "b", "d_r")):
        if c is None:
            assert a is None or float(a.abs().max()) == 0.0, name
            continue
        scale = max(c.abs().max().item(), 1e-20)
        # geometric-consistency tap gradients are float atomics: order-dependent in the last bits
        assert (a - c).abs().max().item() <= 2e-5 * scale + 1e-9, (name, (a - c).abs().max().item(), scale)


def _objective_run(fn, d, d_r, **args):
    leaves = [d[k].clone().requires_grad_(True) for k in ("gt_depth", "gt_pose", "gt_a", "gt_b")]
    dr = None if d_r is None else d_r.clone().requires_grad_(True)
    loss = fn(d["tgt"], d["ref"], leaves[0], dr, leaves[1], d["K"], leaves[2], leaves[3], **args)
    grads = torch.autograd.grad(loss * 2.0, leaves + ([dr] if dr is not None else []), allow_unused=True)
    return loss, grads


@pytest.mark.parametrize("B,H,W,geo,smooth,scales", [(1, 4, 4, 0.5, 0.1, 1), (2, 33, 47, 0.5, 0.1, 1), (2, 36, 52, 0.7, 0.0, 3),
                                                     (1, 64, 124, 0.5, 0.2, 2), (3, 40, 72, 0.25, 0.1, 4)])
def test_full_loss_native_against_live_oracle(B, H, W, geo, smooth, scales):
    """The native objective (geometric term inside the level-0 one-pass kernel) on ragged shapes -- odd widths at one scale,
    strips that end inside a wave, four levels -- against the oracle's dcdp_full_loss, value and every gradient."""
    from coivo_amd import functional as Fh
    from oracle import colvo_spec as S
    b = synth.make_batch(B, H, W, seed=300 + H)
    d = to_dev(b)
    d_r = d["gt_depth"] * 1.07 + 0.05
    args = dict(geo_weight=geo, smooth_weight=smooth, num_scales=scales)
    lh, gh = _objective_run(Fh.dcdp_full_loss, d, d_r, **args)
    bo = {k: v.clone() for k, v in b.items()}
    lo, go = _objective_run(S.dcdp_full_loss, bo, bo["gt_depth"] * 1.07 + 0.05, **args)
    assert abs(lh.item() - lo.item()) < 1e-5
    for a, c, name in zip(gh, go, ("d_t", "pose", "a", "b", "d_r")):
        assert_close_frac(a, c, rtol=2e-3, atol_scale=3e-4 if name.startswith("d_") else 3e-3,
                          max_bad_frac=2e-3 if name.startswith("d_") else 0, what=name)


def test_full_loss_is_bit_reproducible_and_optional_reference_depth():
    """The scatter of the geometric term's tap gradients is fixed-point (integer atomics): two runs agree bit for bit, in
    every gradient.  Without the geometric term the reference depth may be omitted and receives no gradient."""
    from coivo_amd import functional as Fh
    B, H, W = 4, 128, 192
    d = to_dev(synth.make_batch(B, H, W, seed=77))
    d_r = d["gt_depth"] * 0.93 + 0.02
    runs = [_objective_run(Fh.dcdp_full_loss, d, d_r) for _ in range(3)]
    for l, g in runs[1:]:
        assert torch.equal(l, runs[0][0])
        for a, c in zip(g, runs[0][1]):
            assert torch.equal(a, c)
    assert float(runs[0][1][4].abs().max()) > 0.0
    l0, g0 = _objective_run(Fh.dcdp_full_loss, d, None, geo_weight=0.0)
    l1, g1 = _objective_run(Fh.dcdp_full_loss, d, d_r, geo_weight=0.0)
    assert torch.equal(l0, l1) and g1[4] is None
    with pytest.raises(ValueError):
        Fh.dcdp_full_loss(d["tgt"], d["ref"], d["gt_depth"], None, d["gt_pose"], d["K"], d["gt_a"], d["gt_b"])


def test_full_loss_terms_view_and_errors():
    from coivo_amd import functional as Fh
    from oracle import colvo_spec as S
    B, H, W = 2, 48, 64
    b = synth.make_batch(B, H, W, seed=5)
    d = to_dev(b)
    d_t = d["gt_depth"].clone().requires_grad_(True)
    d_r = d["gt_depth"] * 1.1
    loss = Fh.dcdp_full_loss(d["tgt"], d["ref"], d_t, d_r, d["gt_pose"], d["K"], d["gt_a"], d["gt_b"])
    terms = Fh.full_objective_terms(loss).cpu()
    geo = S.geometric_consistency_loss(b["gt_depth"], b["gt_depth"] * 1.1, b["gt_pose"], b["K"]).item()
    sm = S.smoothness_loss(b["gt_depth"], b["tgt"]).item()
    ph = S.photometric_loss(b["tgt"], b["ref"], b["gt_depth"], b["gt_pose"], b["K"], b["gt_a"], b["gt_b"]).item()
    assert abs(terms[0].item() - loss.item()) == 0.0
    assert abs(terms[1].item() - geo) < 1e-5 and abs(terms[2].item() - sm) < 1e-5 and abs(terms[4].item() - ph) < 1e-5
    loss.backward()
    with pytest.raises(ValueError):
        Fh.full_objective_terms(loss)                       # the workspace is gone after the backward
    with pytest.raises(ValueError):
        Fh.dcdp_full_loss(d["tgt"][..., :62], d["ref"][..., :62], d_t[..., :62], d_r[..., :62], d["gt_pose"], d["K"], d["gt_a"],
                          d["gt_b"])                        # 62 is not divisible by 4
    with pytest.raises(ValueError):
        Fh.dcdp_full_loss(d["tgt"], d["ref"], d_t, d_r, d["gt_pose"], d["K"], d["gt_a"], d["gt_b"], num_scales=5)
