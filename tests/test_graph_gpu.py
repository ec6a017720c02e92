"""-m gpu: the hipGraph-captured train step reproduces the eager step (same kernels, same order)."""
import pytest
import torch

from coivo_amd import synth
from tests.gpu_util import dev, to_dev

pytestmark = pytest.mark.gpu


def _setup(seed, dtype):
    from coivo_amd import nn as hnn
    from coivo_amd.optim import FusedAdam
    from oracle import colvo_spec as S
    dn_o, pn_o = S.make_models(seed)
    dn, pn = hnn.DepthNet(compute_dtype=dtype), hnn.PoseNet(compute_dtype=dtype)
    dn.load_state_dict(dn_o.state_dict())
    pn.load_state_dict(pn_o.state_dict())
    return dn, pn, FusedAdam([dn, pn], lr=1e-4)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_graphed_step_matches_eager(dtype):
    from coivo_amd import nn as hnn
    from coivo_amd.graph import GraphedTrainStep
    B, H, W, seed = 2, 64, 96, 61
    b = to_dev(synth.make_batch(B, H, W, seed=seed))
    frames = torch.cat([b["tgt"], b["ref"]])
    dn1, pn1, opt1 = _setup(seed, dtype)
    dn2, pn2, opt2 = _setup(seed, dtype)
    step = GraphedTrainStep(dn2, pn2, opt2, B, H, W)
    eager, graphed = [], []
    for _ in range(4):
        opt1.zero_grad()
        loss = hnn.dcdp_forward(dn1, pn1, b["tgt"], b["ref"], b["K"])[0]
        loss.backward()
        opt1.step()
        eager.append(loss.item())
        graphed.append(step(frames, b["K"]).item())
    # weight gradients use float atomics (order-dependent), so later steps agree to round-off, not bitwise
    tol = 2e-6 if dtype == torch.float32 else 2e-4
    assert abs(eager[0] - graphed[0]) < 1e-7 + tol
    for e, g in zip(eager, graphed):
        assert abs(e - g) < tol * 50, (eager, graphed)
    assert graphed[-1] < graphed[0]
    assert int(opt2.state[0]["step"].item()) == 4     # capture warm-up left no trace in the optimizer state
    # new inputs through the static buffers
    b2 = to_dev(synth.make_batch(B, H, W, seed=seed + 1))
    l_new = step(torch.cat([b2["tgt"], b2["ref"]]), b2["K"]).item()
    assert abs(l_new - graphed[-1]) > 1e-6
