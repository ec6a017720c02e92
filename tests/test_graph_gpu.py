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


@pytest.mark.parametrize("policy", [0, 1, 2, 3])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_graphed_step_matches_eager(dtype, policy):
    from coivo_amd import nn as hnn
    from coivo_amd.graph import GraphedTrainStep
    B, H, W, seed = 2, 64, 96, 61
    b = to_dev(synth.make_batch(B, H, W, seed=seed))
    frames = torch.cat([b["tgt"], b["ref"]])
    dn1, pn1, opt1 = _setup(seed, dtype)
    dn2, pn2, opt2 = _setup(seed, dtype)
    step = GraphedTrainStep(dn2, pn2, opt2, B, H, W, capture_policy=policy, capture_group=5)
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


def test_graphed_step_is_bitwise_the_eager_step_in_deterministic_mode():
    """With deterministic weight gradients nothing in the step depends on an execution order: three replayed steps and three eager
    steps from the same start end in bit-identical parameters, whatever the graph's branch structure."""
    from coivo_amd import nn as hnn
    from coivo_amd.graph import GraphedTrainStep
    B, H, W, seed = 2, 64, 96, 64
    b = to_dev(synth.make_batch(B, H, W, seed=seed))
    frames = torch.cat([b["tgt"], b["ref"]])
    dn1, pn1, opt1 = _setup(seed, torch.bfloat16)
    dn2, pn2, opt2 = _setup(seed, torch.bfloat16)
    for n in (dn1, pn1, dn2, pn2):
        n.deterministic = True
    step = GraphedTrainStep(dn2, pn2, opt2, B, H, W, capture_policy=1)
    for _ in range(3):
        opt1.zero_grad()
        l1 = hnn.dcdp_forward(dn1, pn1, b["tgt"], b["ref"], b["K"])[0]
        l1.backward()
        opt1.step()
        l2 = step(frames, b["K"])
        assert l1.item() == l2.item()
    torch.cuda.synchronize()
    assert torch.equal(dn1.flat_param, dn2.flat_param) and torch.equal(pn1.flat_param, pn2.flat_param)
    assert torch.equal(dn1.flat_grad, dn2.flat_grad) and torch.equal(pn1.flat_grad, pn2.flat_grad)


def test_graphed_full_objective_step_is_bitwise_the_eager_one():
    """The widened objective inside the captured step: its scatter is fixed-point, so in deterministic mode the replayed and the
    eager trajectory agree bit for bit here too."""
    from coivo_amd import nn as hnn
    from coivo_amd.graph import GraphedTrainStep
    B, H, W, seed = 2, 64, 96, 66
    b = to_dev(synth.make_batch(B, H, W, seed=seed))
    frames = torch.cat([b["tgt"], b["ref"]])
    dn1, pn1, opt1 = _setup(seed, torch.bfloat16)
    dn2, pn2, opt2 = _setup(seed, torch.bfloat16)
    for n in (dn1, pn1, dn2, pn2):
        n.deterministic = True
    step = GraphedTrainStep(dn2, pn2, opt2, B, H, W, full_loss=True)
    for _ in range(3):
        opt1.zero_grad()
        l1 = hnn.dcdp_forward(dn1, pn1, b["tgt"], b["ref"], b["K"], full_loss=True)[0]
        l1.backward()
        opt1.step()
        l2 = step(frames, b["K"])
        assert l1.item() == l2.item()
    torch.cuda.synchronize()
    assert torch.equal(dn1.flat_param, dn2.flat_param) and torch.equal(pn1.flat_param, pn2.flat_param)


def test_graph_is_built_with_explicit_dependencies():
    """The captured graph has ONE root (the capture is one stream) and more than one branch: under capture colvo_run_commands
    hangs the weight-gradient chain off the main chain by dependency edits, not by a second captured stream."""
    from coivo_amd.graph import GraphedTrainStep
    B, H, W, seed = 1, 32, 64, 62
    b = to_dev(synth.make_batch(B, H, W, seed=seed))
    dn, pn, opt = _setup(seed, torch.bfloat16)
    step = GraphedTrainStep(dn, pn, opt, B, H, W, capture_policy=2, capture_group=4)
    l0 = step(torch.cat([b["tgt"], b["ref"]]), b["K"]).item()
    l1 = step().item()
    assert 0 < l1 < 1 and l1 != l0
    assert dn._side is None or not torch.cuda.is_current_stream_capturing()


def test_graphed_step_with_rccl_single_rank():
    """configs[4]'s structure on one GPU: the graph contains the bucketed all-reduces of ddp.GradBuckets (RCCL, one rank) beside the
    native two-chain backward, and replays to the same losses as the eager data-parallel step."""
    import os
    import torch.distributed as dist
    from coivo_amd import nn as hnn
    from coivo_amd.ddp import GradBuckets
    from coivo_amd.graph import GraphedTrainStep
    if dist.is_initialized():
        pytest.skip("a process group is already initialised in this process")
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev())
    try:
        B, H, W, seed = 2, 64, 96, 63
        b = to_dev(synth.make_batch(B, H, W, seed=seed))
        frames = torch.cat([b["tgt"], b["ref"]])
        dn1, pn1, opt1 = _setup(seed, torch.bfloat16)
        dn2, pn2, opt2 = _setup(seed, torch.bfloat16)
        ddp1 = GradBuckets([dn1, pn1], bucket_bytes=4 << 20, transport_dtype=torch.bfloat16)
        ddp2 = GradBuckets([dn2, pn2], bucket_bytes=4 << 20, transport_dtype=torch.bfloat16)
        step = GraphedTrainStep(dn2, pn2, opt2, B, H, W, ddp=ddp2)
        eager, graphed = [], []
        for _ in range(3):
            opt1.zero_grad()
            loss = hnn.dcdp_forward(dn1, pn1, b["tgt"], b["ref"], b["K"])[0]
            loss.backward()
            ddp1.finish()
            opt1.step()
            eager.append(loss.item())
            graphed.append(step(frames, b["K"]).item())
        torch.cuda.synchronize()
        for e, g in zip(eager, graphed):
            assert abs(e - g) < 1e-2, (eager, graphed)
        assert graphed[-1] < graphed[0]
        ddp1.detach(); ddp2.detach()
    finally:
        dist.destroy_process_group()
