"""World-size-2 gloo tests of the data-parallel gradient exchange (coivo_amd.ddp.GradBuckets) on CPU.

The HIP networks cannot run here (no GPU), so the arena protocol is driven by a small CPU stand-in that wraps the
ORACLE networks: flat gradient arena, layers reported in reverse arena order -- exactly what coivo_amd.nn does.
Checked: (1) bucket boundaries / ordering / deferred layers reduce every element exactly once;
(2) 2-rank data parallel with per-rank batches == single-process big batch (BN-free spec), through Adam.
"""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


class FakeArena:
    """Minimal object with the arena interface GradBuckets needs."""

    def __init__(self, n):
        self.flat_grad = torch.zeros(n)
        self.grad_ready_hook = None


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _init(rank, world, port):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)


def _worker_buckets(rank, world, port, transport):
    from coivo_amd.ddp import GradBuckets
    _init(rank, world, port)
    torch.manual_seed(100 + rank)
    n = 1000
    spans = [(0, 130), (130, 400), (400, 410), (410, 777), (777, 1000)]      # layers, arena order
    a, b = FakeArena(n), FakeArena(37)
    gb = GradBuckets([a, b], bucket_bytes=4 * 256, transport_dtype=transport)
    for step in range(3):
        ga, gbv = torch.randn(n), torch.randn(37)
        a.flat_grad.copy_(ga)
        b.flat_grad.copy_(gbv)
        order = list(reversed(spans))
        if step == 1:                      # a layer reports out of order -> deferred to finish()
            order[1], order[2] = order[2], order[1]
        if step == 2:
            order = order[:-2]             # some layers never report -> finish() picks them up
        for lo, hi in order:
            a.grad_ready_hook(a, lo, hi)
        b.grad_ready_hook(b, 0, 37)
        gb.finish()
        ref_a = [torch.zeros(n) for _ in range(world)]
        dist.all_gather(ref_a, ga)
        ref_b = [torch.zeros(37) for _ in range(world)]
        dist.all_gather(ref_b, gbv)
        # fp32 transport: two ranks add commutatively (exact); more ranks are summed in the collective's own order (rounding)
        tol = (dict(rtol=0, atol=0 if world == 2 else 1e-5) if transport is None else dict(rtol=2e-2, atol=1e-2 * world))
        assert torch.allclose(a.flat_grad, sum(ref_a), **tol), f"step {step}"
        assert torch.allclose(b.flat_grad, sum(ref_b), **tol), f"step {step}"
    assert abs(gb.grad_scale - 1.0 / world) < 1e-12
    dist.destroy_process_group()


@pytest.mark.parametrize("transport", [None, torch.bfloat16])
def test_bucketed_allreduce_world2(transport):
    mp.spawn(_worker_buckets, args=(2, _free_port(), transport), nprocs=2, join=True)


@pytest.mark.parametrize("transport", [None, torch.bfloat16])
def test_bucketed_allreduce_world8(transport):
    """The same exchange at the rank count of BASELINE configs[3] / [4] (8 GPUs): bucket order, deferred and missing layers, the
    1/8 gradient scale; the only rehearsal of N = 8 this build can run."""
    mp.spawn(_worker_buckets, args=(8, _free_port(), transport), nprocs=8, join=True)


class OracleArena:
    """Wraps the oracle networks behind the arena protocol (CPU stand-in for coivo_amd.nn._ArenaModule)."""

    def __init__(self, net):
        self.net = net
        self.params = list(net.parameters())
        self.flat_grad = torch.zeros(sum(p.numel() for p in self.params))
        self.grad_ready_hook = None

    def publish_grads(self):
        off, spans = 0, []
        for p in self.params:
            self.flat_grad[off:off + p.numel()].copy_(p.grad.reshape(-1))
            spans.append((off, off + p.numel()))
            off += p.numel()
        for lo, hi in reversed(spans):          # backward finishes the last layer first
            self.grad_ready_hook(self, lo, hi)

    def scatter_back(self, scale):
        off = 0
        for p in self.params:
            p.grad.copy_((self.flat_grad[off:off + p.numel()] * scale).view_as(p))
            off += p.numel()


def _worker_dp(rank, world, port, out):
    from coivo_amd import synth
    from coivo_amd.ddp import GradBuckets
    from oracle import colvo_spec as S
    _init(rank, world, port)
    torch.set_num_threads(2)
    B, H, W = 2, 32, 64
    full = synth.make_batch(B, H, W, seed=7)
    sl = slice(rank, rank + 1)                         # one frame pair per rank
    dn, pn = S.make_models(3)
    arenas = [OracleArena(dn), OracleArena(pn)]
    gb = GradBuckets(arenas, bucket_bytes=1 << 20)
    opt = torch.optim.Adam(list(dn.parameters()) + list(pn.parameters()), **S.ADAM_KW)
    for _ in range(2):
        opt.zero_grad()
        loss = S.dcdp_forward(dn, pn, full["tgt"][sl], full["ref"][sl], full["K"][sl])[0]
        loss.backward()
        for a in arenas:
            a.publish_grads()
        gb.finish()
        for a in arenas:
            a.scatter_back(gb.grad_scale)
        opt.step()
    if rank == 0:
        torch.save({k: v.clone() for k, v in dn.state_dict().items()}, out)
    dist.destroy_process_group()


def test_two_rank_dp_matches_mean_of_per_rank_losses(tmp_path):
    """DP semantics: the update equals single-process training on the mean of the per-rank losses."""
    from coivo_amd import synth
    from oracle import colvo_spec as S
    out = str(tmp_path / "dn_rank0.pt")
    mp.spawn(_worker_dp, args=(2, _free_port(), out), nprocs=2, join=True)
    got = torch.load(out)
    torch.set_num_threads(2)
    full = synth.make_batch(2, 32, 64, seed=7)
    dn, pn = S.make_models(3)
    opt = torch.optim.Adam(list(dn.parameters()) + list(pn.parameters()), **S.ADAM_KW)
    for _ in range(2):
        opt.zero_grad()
        losses = [S.dcdp_forward(dn, pn, full["tgt"][i:i + 1], full["ref"][i:i + 1], full["K"][i:i + 1])[0] for i in (0, 1)]
        (sum(losses) / 2).backward()
        opt.step()
    for k, v in dn.state_dict().items():
        assert torch.allclose(got[k], v, rtol=1e-4, atol=1e-6), k


def test_gradbuckets_needs_process_group():
    from coivo_amd.ddp import GradBuckets
    if dist.is_initialized():
        pytest.skip("a process group is active in this process")
    with pytest.raises(RuntimeError):
        GradBuckets([FakeArena(8)])


def _worker_rules(rank, world, port):
    """Round 6 rules that need no GPU: the deferred loss normalisation needs its optimizer; a captured step on an nccl group is refused
    through ProcessGroup.allreduce (the backend name is monkeypatched: the decision, not the capture, is what runs here)."""
    from coivo_amd import graph
    from coivo_amd.ddp import GradBuckets
    _init(rank, world, port)
    try:
        a = FakeArena(1000)
        with pytest.raises(ValueError, match="optimizer"):
            GradBuckets([a], defer_loss_normalisation=True)
        ddp = GradBuckets([a])
        assert ddp.native_collectives is False and ddp.native_fallback is None and ddp._defer is False

        class Step:             # the two attributes _process_group_path reads
            allow_process_group_capture = False
        st = Step()
        st.ddp = ddp
        assert graph.GraphedTrainStep._process_group_path(st) is False           # gloo: nothing to guard
        real = dist.get_backend
        dist.get_backend = lambda group=None: "nccl"
        try:
            with pytest.raises(RuntimeError, match="native_collectives=True"):
                graph.GraphedTrainStep._process_group_path(st)
            st.allow_process_group_capture = True
            assert graph.GraphedTrainStep._process_group_path(st) is True        # opted in: capture() runs the timing-based guard
            ddp._native = object()                                               # native path: no guard at all
            st.allow_process_group_capture = False
            assert graph.GraphedTrainStep._process_group_path(st) is False
            ddp._native = None
        finally:
            dist.get_backend = real
        ddp.drain_eager_collectives()                                            # nothing issued: a no-op
        ddp.detach()
    finally:
        dist.destroy_process_group()


def test_round6_rules_for_deferred_normalisation_and_captured_nccl_steps():
    mp.spawn(_worker_rules, args=(1, _free_port()), nprocs=1, join=True)
