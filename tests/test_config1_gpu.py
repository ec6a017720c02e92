"""-m gpu: the configuration bench.py measures -- BASELINE configs[1]: batch 8, 320x256 (tensors [8,3,256,320]), i.e.
DepthNet on 16 images -- checked against the oracle.  The conv dispatch (tile widths, persistent / ring kernels, weight
gradient splits) depends on the grid size, so the kernel variants of THIS shape are the ones covered here.

fp32 mode: loss 1e-5, depth 1e-4 (BASELINE.json north_star), every parameter gradient at SPEC.md §7's bar with the fp64
oracle as the yardstick.  bf16 mode (the bench's dtype): bounds on 'depth L1 vs ref' and on the loss.
"""
import os

import pytest
import torch

from coivo_amd import synth
from tests.gpu_util import dev, grad_parity_failures, matched_grad_rows, oracle_step as _oracle_step, to_dev

pytestmark = pytest.mark.gpu

B, H, W, SEED = 8, 256, 320, 1234       # bench.py's batch: synth.make_batch(8, 256, 320, seed=1234)
DEPTH_TOL, LOSS_TOL = 1e-4, 1e-5
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _models(dtype):
    from coivo_amd import nn as hnn
    from oracle import colvo_spec as S
    dn_o, pn_o = S.make_models(0)
    dn, pn = hnn.DepthNet(compute_dtype=dtype), hnn.PoseNet(compute_dtype=dtype)
    dn.load_state_dict(dn_o.state_dict())
    pn.load_state_dict(pn_o.state_dict())
    return dn_o, pn_o, dn, pn


@pytest.fixture(scope="module")
def oracle_step():
    """The fp32 oracle's coupled step on the bench batch (weights: spec init, seed 0)."""
    b = synth.make_batch(B, H, W, seed=SEED)
    return b, {"f32": _oracle_step(SEED, b, torch.float32, weights_seed=0)}


def test_config1_fp32_step_parity(oracle_step):
    from coivo_amd import nn as hnn
    b, o = oracle_step
    _, _, dn, pn = _models(torch.float32)
    d = to_dev(b)
    loss, d_t, d_r, pose, a, bb = hnn.dcdp_forward(dn, pn, d["tgt"], d["ref"], d["K"])
    loss.backward()
    torch.cuda.synchronize()
    o32 = o["f32"]
    assert abs(loss.item() - o32["loss"]) < LOSS_TOL, (loss.item(), o32["loss"])
    assert (d_t.detach().cpu() - o32["d_t"]).abs().max().item() < DEPTH_TOL
    assert (d_r.detach().cpu() - o32["d_r"]).abs().max().item() < DEPTH_TOL
    assert (pose.detach().cpu() - o32["pose"]).abs().max().item() < 1e-6
    assert (a.detach().cpu() - o32["a"]).abs().max().item() < 1e-6
    assert (bb.detach().cpu() - o32["b"]).abs().max().item() < 1e-6
    # every parameter gradient: fp64 oracle as the yardstick, fp32 oracle as the noise scale, both evaluated at the HIP path's
    # ReLU decisions (at this size hundreds of pre-activations lie within rounding distance of zero: tests/gpu_util.py)
    out = os.path.join(ROOT, "gpurun_out", "grad_parity_config1.txt") if os.path.isdir(os.path.join(ROOT, "gpurun_out")) else None
    rows, m32, _ = matched_grad_rows(SEED, b, dn, pn, out, weights_seed=0)
    assert len(rows) == 58
    assert abs(loss.item() - m32["loss"]) < LOSS_TOL
    bad = grad_parity_failures(rows)
    assert not bad, "\n".join(bad)


def test_config1_bf16_depth_l1_and_loss(oracle_step):
    """The bench's compute dtype on the bench's batch: 'depth L1 vs ref' (BASELINE.json metric) and the loss stay within
    bf16-sized bounds of the fp32 oracle; gradients point the same way."""
    from coivo_amd import nn as hnn
    b, o = oracle_step
    _, _, dn, pn = _models(torch.bfloat16)
    d = to_dev(b)
    loss, d_t, d_r = hnn.dcdp_forward(dn, pn, d["tgt"], d["ref"], d["K"])[:3]
    loss.backward()
    o32 = o["f32"]
    l1 = 0.5 * ((d_t.detach().cpu() - o32["d_t"]).abs().mean().item() + (d_r.detach().cpu() - o32["d_r"]).abs().mean().item())
    rel = l1 / o32["d_t"].abs().mean().item()
    line = f"configs[1] bf16: depth L1 vs ref {l1:.3e} (relative {rel:.3e}), loss {loss.item():.6f} vs {o32['loss']:.6f}"
    print(line)
    if os.path.isdir(os.path.join(ROOT, "gpurun_out")):
        with open(os.path.join(ROOT, "gpurun_out", "config1_bf16.txt"), "w") as f:
            f.write(line + "\n")
    # about 3 x what the bench line reports for this batch ('depth_l1_vs_oracle': mean_rel 1.1e-3 in rounds 3-4); round 4 allowed 1e-2
    assert rel < 3.5e-3
    assert abs(loss.item() - o32["loss"]) < 3e-4          # (observed 4.4e-5 on a loss of 0.049; round 4 allowed 2e-3)
    og = dict(o32["grads"])
    for net, tag, names in ((dn, "depth.", ("enc3b.weight", "iconv3.weight", "up1.weight", "head.weight")),
                            (pn, "pose.", ("conv2.weight", "conv6.weight"))):
        for n in names:
            gh = dict(net.named_parameters())[n].grad.float().cpu().flatten()
            cos = torch.nn.functional.cosine_similarity(gh, og[tag + n].flatten(), dim=0).item()
            assert cos > 0.97, (tag + n, cos)
    assert torch.isfinite(dn.flat_grad).all() and torch.isfinite(pn.flat_grad).all()
