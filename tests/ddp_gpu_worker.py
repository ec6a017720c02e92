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


def _check_against_oracle(rank, world, seed, B, full, dn, pn, loss, opt, g_dn, g_pn):
    """DP == big batch against the SPEC, un-redefined: the all-reduced gradient x 1/world and the parameters
    after one Adam step vs the oracle's single-process step on the WHOLE batch of world x B pairs -- ONE masked mean over all of its
    valid pixels (oracle/SPEC.md section 5; GradBuckets(exact_batch_loss=True) all-reduces the valid-pixel count and the masked sum
    behind the loss kernel) -- fp32 and fp64 oracle at the HIP path's ReLU decisions of every rank's slice (tests/gpu_util.py), at
    the bar of grad_parity_failures.  Rounds 3-4 compared with the mean of the per-rank losses instead.  Every rank must report
    the loss of the whole batch."""
    from oracle import colvo_spec as S
    from tests import gpu_util as G
    masks = G.hip_relu_masks(dn, pn)
    gathered = [None] * world if rank == 0 else None
    dist.gather_object((masks, float(loss.item())), gathered, dst=0)
    if rank != 0:
        return
    cpu = {k: v.detach().cpu() for k, v in full.items() if torch.is_tensor(v)}
    # the ranks' ReLU decisions in the big batch's order: DepthNet runs on [all target frames | all reference frames]
    big = {}
    for key in gathered[0][0]:
        parts = [g[0][key] for g in gathered]
        if key.startswith("depth."):
            big[key] = torch.cat([p[:B] for p in parts] + [p[B:] for p in parts], dim=0)
        else:
            big[key] = torch.cat(parts, dim=0)
    acc = {}
    for dtype in (torch.float32, torch.float64):
        o = G.oracle_step(seed, {k: cpu[k] for k in ("tgt", "ref", "K")}, dtype, big)
        assert o["flip_worst"] < G.RELU_MARGIN, f"forced a ReLU decision at |pre| = {o['flip_worst']:.3e}"
        acc[dtype] = ([(n, t.detach().double()) for n, t in o["grads"]], o["loss"])
    assert all(abs(l - gathered[0][1]) < 1e-7 for _, l in gathered), [l for _, l in gathered]     # every rank: the batch's loss
    hip_loss = gathered[0][1]
    assert abs(hip_loss - acc[torch.float32][1]) < 1e-5, (hip_loss, acc[torch.float32][1])
    # (the per-rank means differ from it: this batch would not pass with the round-4 normalisation)
    per_rank = [G.oracle_step(seed, {k: cpu[k][r * B:(r + 1) * B] for k in ("tgt", "ref", "K")}, torch.float32)["loss"] for r in range(world)]
    print(f"oracle: batch loss {acc[torch.float32][1]:.7f}, mean of the per-rank losses {sum(per_rank) / world:.7f}", flush=True)
    # (DDP_DEFER=1: the arena holds the sum of the ranks' UNNORMALISED gradients and the optimizer has been handed the device-side
    #  factor world / max(3 n_global, 1); the gradient of the batch loss is their product / world)
    gs = (float(opt.grad_scale_dev.item()) if opt.grad_scale_dev is not None else 1.0) / world
    assert (opt.grad_scale_dev is not None) == (os.environ.get("DDP_DEFER") is not None)
    hip = [("depth." + n, p.grad.detach().double() * gs) for n, p in dn.named_parameters()] + \
          [("pose." + n, p.grad.detach().double() * gs) for n, p in pn.named_parameters()]
    rows = G.grad_parity_table(hip, acc[torch.float32][0], acc[torch.float64][0])
    bad = G.grad_parity_failures(rows)
    assert not bad, "data-parallel gradient vs the oracle's gradient of the whole batch's loss:\n" + "\n".join(bad)
    # one Adam step of the oracle on that gradient vs FusedAdam(grad_scale = 1/world) on the all-reduced arena
    dn_o, pn_o = S.make_models(seed)
    params = list(dn_o.parameters()) + list(pn_o.parameters())
    for p, (_, g) in zip(params, acc[torch.float32][0]):
        p.grad = g.float()
    torch.optim.Adam(params, **S.ADAM_KW).step()
    dn2, pn2 = type(dn)(device=dn.flat_param.device), type(pn)(device=pn.flat_param.device)
    dn2.load_state_dict(dn.state_dict()); pn2.load_state_dict(pn.state_dict())     # copies: the real step follows in main()
    dn2.flat_grad.copy_(dn.flat_grad); pn2.flat_grad.copy_(pn.flat_grad)
    from coivo_amd.optim import FusedAdam
    opt2 = FusedAdam([dn2, pn2], lr=opt.lr)
    opt2.grad_scale = opt.grad_scale
    opt2.grad_scale_dev = opt.grad_scale_dev
    dn2.attach_grads(); pn2.attach_grads()
    dn2.flat_grad.copy_(dn.flat_grad); pn2.flat_grad.copy_(pn.flat_grad)
    opt2.step()
    torch.cuda.synchronize()
    worst, flipped, total = 0.0, 0, 0
    hp = dict([("depth." + n, p) for n, p in dn2.named_parameters()] + [("pose." + n, p) for n, p in pn2.named_parameters()])
    op = dict([("depth." + n, p) for n, p in dn_o.named_parameters()] + [("pose." + n, p) for n, p in pn_o.named_parameters()])
    for n in hp:
        d = (hp[n].detach().cpu() - op[n].detach()).abs()
        worst = max(worst, d.max().item())
        flipped += int((d > 1e-6).sum()); total += d.numel()
    # Adam's first step is sign-like (+-lr whatever the gradient's size): only elements whose gradient is rounding noise around zero
    # may land on the other side (2 lr apart); everything else must agree to float rounding
    assert worst <= 2.0 * opt.lr * 1.01 + 1e-6 and flipped / total < 5e-3, (worst, flipped, total)
    print(f"DDP_OK ORACLE: loss |d| {abs(hip_loss - acc[torch.float32][1]):.1e}, worst gradient tensor relL2 "
          f"{max(r[5] for r in rows):.2e} (fp32 oracle {max(r[6] for r in rows):.2e}), parameters after Adam: max |d| {worst:.2e}, "
          f"{flipped}/{total} elements beyond 1e-6", flush=True)


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
    # DDP_DEFER=1: the loss normaliser's exchange overlaps the backward pass and the global scale goes into the optimizer (round 6)
    defer = os.environ.get("DDP_DEFER") is not None and not obj
    ddp = GradBuckets([dn, pn], bucket_bytes=2 << 20,          # several buckets per arena
                      **(dict(defer_loss_normalisation=True, optimizer=opt) if defer else {}))
    opt.grad_scale = ddp.grad_scale
    opt.zero_grad()
    loss = hnn.dcdp_forward(dn, pn, full["tgt"][sl], full["ref"][sl], full["K"][sl], **obj)[0]
    loss.backward()
    ddp.finish()
    torch.cuda.synchronize()
    g_dn, g_pn = dn.flat_grad.clone(), pn.flat_grad.clone()       # summed over ranks, in place in the arena
    if os.environ.get("DDP_ORACLE") is not None:
        _check_against_oracle(rank, world, seed, B, full, dn, pn, loss, opt, g_dn, g_pn)
    opt.step()
    torch.cuda.synchronize()
    # every rank must hold identical gradients and parameters
    for t in (g_dn, dn.flat_param):
        ref = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(ref, t)
        assert all(torch.equal(r, ref[0]) for r in ref), "ranks diverged"
    if rank == 0:
        # single-process reference: the per-rank batches accumulated into one arena -- with the normalisation the ranks used: the
        # photometric loss of every slice takes the WHOLE batch's valid-pixel count and masked sum (what the ranks all-reduce behind
        # their loss kernels), here added up from a forward-only pass over the slices; the widened objective has no such exchange
        from coivo_amd import _lib, functional as Fh
        dn2, pn2 = fresh()
        opt2 = FusedAdam([dn2, pn2], lr=1e-4)
        opt2.grad_scale = 1.0 / world
        slices = [slice(r * B, (r + 1) * B) for r in range(world)]
        if not obj:
            seen = []
            Fh.set_batch_reducer(lambda st, can_defer=False: seen.append(st.clone()))
            for s2 in slices:
                hnn.dcdp_forward(dn2, pn2, full["tgt"][s2], full["ref"][s2], full["K"][s2])
            assert len(seen) == world
            glob = seen[0][2:4].clone()
            for st in seen[1:]:
                glob += st[2:4]

            one_ref, norm_ref = torch.ones(1, device=dev), torch.ones(1, device=dev)

            def whole_batch(st, can_defer=False):
                st[2:4].copy_(glob)
                if defer and can_defer:     # what the ranks did: backward on the raw gradients, the global scale through the optimizer
                    _lib.check(_lib.load().colvo_warp_loss_rescale_to(_lib.ptr(st), world, _lib.ptr(norm_ref), _lib.stream_ptr()),
                               "colvo_warp_loss_rescale_to")
                    opt2.grad_scale_dev = norm_ref
                    return one_ref
                assert not defer
                _lib.check(_lib.load().colvo_warp_loss_rescale(_lib.ptr(st), world, _lib.stream_ptr()), "colvo_warp_loss_rescale")
                return None
            Fh.set_batch_reducer(whole_batch)
        opt2.zero_grad()
        for s2 in slices:
            hnn.dcdp_forward(dn2, pn2, full["tgt"][s2], full["ref"][s2], full["K"][s2], **obj)[0].backward()
        ddp._install_reducer(True)               # back to the data-parallel one for the steps below
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
        print("DDP_OK" + (" DET" if det else "") + (" FULL" if obj else "") + (" DEFER" if defer else ""), f"command-list calls: {n_learned} (learned) vs {n_full} (per layer)", flush=True)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
