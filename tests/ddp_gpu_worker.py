"""Worker for tests/test_ddp_gpu.py: launched by torch.distributed.run with 2 ranks that SHARE cuda:0 (the test box has one
GPU; RCCL cannot put two ranks on one device, so the rehearsal carries the CUDA tensors over gloo).  Exercises the real
HIP networks: backward hooks in true layer order, weight gradients on the side stream, bucketed async all-reduce,
finish(), grad_scale in the fused Adam."""
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from coivo_amd import nn as hnn
    from coivo_amd import synth
    from coivo_amd.ddp import GradBuckets
    from coivo_amd.optim import FusedAdam
    from oracle import colvo_spec as S
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    B, H, W, seed = 1, 64, 96, 71
    det = os.environ.get("DDP_DET") is not None
    obj = dict(full_loss=True) if os.environ.get("DDP_FULL") is not None else {}      # DDP_FULL=1: the widened objective
    dn_o, pn_o = S.make_models(seed)

    def fresh():
        dn, pn = hnn.DepthNet(device=dev), hnn.PoseNet(device=dev)
        dn.load_state_dict(dn_o.state_dict())
        pn.load_state_dict(pn_o.state_dict())
        dn.deterministic = pn.deterministic = det        # DDP_DET=1: weight gradients without float atomics -> exact comparisons
        return dn, pn

    full = synth.make_batch(world * B, H, W, seed=seed, device=dev)
    sl = slice(rank * B, (rank + 1) * B)
    dn, pn = fresh()
    opt = FusedAdam([dn, pn], lr=1e-4)
    ddp = GradBuckets([dn, pn], bucket_bytes=2 << 20)          # several buckets per arena
    opt.grad_scale = ddp.grad_scale
    opt.zero_grad()
    loss = hnn.dcdp_forward(dn, pn, full["tgt"][sl], full["ref"][sl], full["K"][sl], **obj)[0]
    loss.backward()
    ddp.finish()
    torch.cuda.synchronize()
    g_dn, g_pn = dn.flat_grad.clone(), pn.flat_grad.clone()       # summed over ranks, in place in the arena
    opt.step()
    torch.cuda.synchronize()
    # every rank must hold identical gradients and parameters
    for t in (g_dn, dn.flat_param):
        ref = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(ref, t)
        assert all(torch.equal(r, ref[0]) for r in ref), "ranks diverged"
    if rank == 0:
        # single-process reference: the per-rank batches accumulated into one arena
        dn2, pn2 = fresh()
        opt2 = FusedAdam([dn2, pn2], lr=1e-4)
        opt2.grad_scale = 1.0 / world
        opt2.zero_grad()
        for r in range(world):
            s2 = slice(r * B, (r + 1) * B)
            hnn.dcdp_forward(dn2, pn2, full["tgt"][s2], full["ref"][s2], full["K"][s2], **obj)[0].backward()
        torch.cuda.synchronize()
        for a, b, name in ((g_dn, dn2.flat_grad, "DepthNet"), (g_pn, pn2.flat_grad, "PoseNet")):
            scale = b.abs().max().item()
            err = (a - b).abs().max().item()
            assert err < 2e-4 * scale, f"{name}: all-reduced gradient differs from the accumulated reference by {err} (scale {scale})"
            if det:      # data parallel == one process accumulating the ranks' batches, bit for bit (the spec has no BatchNorm)
                assert torch.equal(a, b), f"{name}: deterministic mode, yet the all-reduced gradient is not bitwise the accumulated one"
        opt2.step()
        torch.cuda.synchronize()
        # Adam's first step is sign-like (+-lr): float-atomic summation order may flip it on ~zero gradients only
        d = (dn.flat_param - dn2.flat_param).abs()
        assert d.max().item() <= 2.5e-4 and (d > 1e-6).float().mean().item() < 5e-3
        if det:
            assert torch.equal(dn.flat_param, dn2.flat_param) and torch.equal(pn.flat_param, pn2.flat_param)
    # second backward from the same parameters, twice: with the split points learned during the first step (few segments) and
    # with the learning pass forced again (one segment per layer) -- same gradients, fewer command-list calls
    from coivo_amd.program import Program
    calls = [0]
    orig_run = Program.run

    def counting_run(self, *a, **kw):
        calls[0] += 1
        return orig_run(self, *a, **kw)

    Program.run = counting_run

    def ddp_backward():
        calls[0] = 0
        opt.zero_grad()
        hnn.dcdp_forward(dn, pn, full["tgt"][sl], full["ref"][sl], full["K"][sl], **obj)[0].backward()
        ddp.finish()
        torch.cuda.synchronize()
        return dn.flat_grad.clone(), pn.flat_grad.clone(), calls[0]

    ga_dn, ga_pn, n_learned = ddp_backward()
    for net in (dn, pn):
        for insts in net._insts.values():
            for inst in insts:
                for pr, _ in inst.passes.values():
                    pr.flush = None
    gb_dn, gb_pn, n_full = ddp_backward()
    Program.run = orig_run
    assert n_learned < n_full, (n_learned, n_full)
    for a, b, name in ((ga_dn, gb_dn, "DepthNet"), (ga_pn, gb_pn, "PoseNet")):
        scale = b.abs().max().item()
        assert (a - b).abs().max().item() < 2e-4 * scale, f"{name}: learned split points changed the all-reduced gradient"
        if det:
            assert torch.equal(a, b), f"{name}: deterministic mode, yet the learned split points changed the gradient bits"
    if rank == 0:
        print("DDP_OK" + (" DET" if det else "") + (" FULL" if obj else ""), f"command-list calls: {n_learned} (learned) vs {n_full} (per layer)", flush=True)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
