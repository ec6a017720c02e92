"""-m gpu: the recorded command lists (coivo_amd/program.py, colvo_run_commands) behave like the layer-by-layer path.

Covers what recording adds on top of the kernels: persistent activation buffers, per-call pointer patching, pass
instances leased from forward to backward, and re-recording after the operand copies move.
"""
import pytest
import torch

from coivo_amd import synth
from tests.gpu_util import dev, to_dev

pytestmark = pytest.mark.gpu


def _nets(seed=5, dtype=torch.float32, programs=True):
    from coivo_amd import nn as hnn
    from oracle import colvo_spec as S
    dn_o, pn_o = S.make_models(seed)
    dn, pn = hnn.DepthNet(compute_dtype=dtype), hnn.PoseNet(compute_dtype=dtype)
    dn.load_state_dict(dn_o.state_dict())
    pn.load_state_dict(pn_o.state_dict())
    dn.use_programs = pn.use_programs = programs
    return dn, pn


def _step(dn, pn, b):
    from coivo_amd import nn as hnn
    dn.zero_grad(); pn.zero_grad()
    loss, d_t, d_r, pose, a, bb = hnn.dcdp_forward(dn, pn, b["tgt"], b["ref"], b["K"])
    loss.backward()
    return loss.detach().clone(), d_t.detach().clone(), dn.flat_grad.clone(), pn.flat_grad.clone()


def _close(a, b, tol, what):
    scale = max(b.abs().max().item(), 1e-30)
    err = (a - b).abs().max().item()
    assert err <= tol * scale, f"{what}: max err {err:.3e} vs scale {scale:.3e}"


def test_replay_equals_layer_by_layer_and_follows_new_inputs():
    B, H, W = 2, 64, 96
    dn_p, pn_p = _nets(programs=True)
    dn_e, pn_e = _nets(programs=False)
    for seed in (31, 32, 33):                     # first call records, the next two replay with patched pointers
        b = to_dev(synth.make_batch(B, H, W, seed=seed))
        lp, dp, gdp, gpp = _step(dn_p, pn_p, b)
        le, de, gde, gpe = _step(dn_e, pn_e, b)
        assert torch.equal(dp, de), "forward kernels are deterministic: replay must be bit-identical"
        assert lp.item() == le.item()
        _close(gdp, gde, 1e-5, "DepthNet gradients")       # weight gradients use float atomics: not bitwise
        _close(gpp, gpe, 1e-5, "PoseNet gradients")
    assert len(dn_p._insts) == 1 and all(len(v) == 1 for v in dn_p._insts.values())


def test_two_forwards_before_backward_get_their_own_buffers():
    B, H, W = 1, 32, 64
    dn, _ = _nets()
    x1 = to_dev(synth.make_batch(B, H, W, seed=41))["tgt"]
    x2 = to_dev(synth.make_batch(B, H, W, seed=42))["tgt"]
    # reference: one at a time
    dn.zero_grad(); dn(x1).square().sum().backward(); g1 = dn.flat_grad.clone()
    dn.zero_grad(); dn(x2).square().sum().backward(); g2 = dn.flat_grad.clone()
    # both forwards first: the second must not overwrite the activations saved by the first
    dn.zero_grad()
    y1, y2 = dn(x1), dn(x2)
    (y1.square().sum() + y2.square().sum()).backward()
    _close(dn.flat_grad, g1 + g2, 1e-5, "sum of the two backward passes")
    key = next(iter(dn._insts))
    assert len(dn._insts[key]) == 2 and not any(i.busy for i in dn._insts[key])


def test_no_grad_forward_releases_its_instance_and_dtype_switch_rerecords():
    B, H, W = 1, 32, 64
    dn, _ = _nets()
    x = to_dev(synth.make_batch(B, H, W, seed=43))["tgt"]
    with torch.no_grad():
        y0 = dn(x).clone()
    assert not any(i.busy for v in dn._insts.values() for i in v)
    y1 = dn(x)
    assert torch.equal(y0, y1.detach())
    y1.sum().backward()
    # in-place parameter update -> operand copies are re-packed in place, the recorded program stays valid
    with torch.no_grad():
        dn.flat_param.mul_(1.01)
    dn.mark_params_changed()
    y2 = dn(x).detach()
    assert not torch.equal(y2, y0)
    # switching the compute dtype re-allocates the operand copies: programs must be dropped and re-recorded
    dn.compute_dtype = torch.bfloat16
    y3 = dn(x).detach()
    assert (y3 - y2).abs().max().item() < 5e-2 and not torch.equal(y3, y2)
    dn.compute_dtype = torch.float32
    y4 = dn(x).detach()
    assert torch.equal(y4, y2)


def test_aliasing_arguments_record_and_replay():
    """pose_net(x, x): the two frame arguments share one address; the recorded pass must not confuse them, and a later
    call with distinct frames must patch both."""
    from coivo_amd import nn as hnn
    from oracle import colvo_spec as S
    _, pn_o = S.make_models(3)
    pn = hnn.PoseNet()
    pn.load_state_dict(pn_o.state_dict())
    b = synth.make_batch(2, 64, 96, seed=3)
    x, y = b["tgt"].to(dev()), b["ref"].to(dev())
    with torch.no_grad():
        po = pn_o(b["tgt"], b["tgt"])[0]
        ph = pn(x, x)[0]
        assert (ph.cpu() - po).abs().max().item() < 1e-6
        po2 = pn_o(b["tgt"], b["ref"])[0]
        ph2 = pn(x, y)[0]
        assert (ph2.cpu() - po2).abs().max().item() < 1e-6
