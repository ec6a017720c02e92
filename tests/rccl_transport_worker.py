"""Worker for tests/test_ddp_gpu.py::test_bf16_gradient_transport_is_exactly_a_bf16_rounding: one rank, RCCL.
With ONE rank an all-reduce is the identity, so after backward + GradBuckets.finish() the gradient arena of a bf16-transport run must
hold EXACTLY the fp32-transport run's gradients rounded to bf16 (and the fp32-transport run exactly the run without a process group):
every bucket went out, came back and was copied to the right place.  Deterministic weight gradients make the three runs comparable
bit for bit."""
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    from coivo_amd import nn as hnn
    from coivo_amd import synth
    from coivo_amd.ddp import GradBuckets
    from coivo_amd.optim import FusedAdam
    from oracle import colvo_spec as S
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29551")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    B, H, W, seed = 2, 64, 96, 67
    b = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in synth.make_batch(B, H, W, seed=seed).items()}

    def grads(transport, with_ddp=True):
        dn_o, pn_o = S.make_models(seed)
        dn, pn = hnn.DepthNet(compute_dtype=torch.bfloat16), hnn.PoseNet(compute_dtype=torch.bfloat16)
        dn.load_state_dict(dn_o.state_dict())
        pn.load_state_dict(pn_o.state_dict())
        dn.deterministic = pn.deterministic = True
        opt = FusedAdam([dn, pn], lr=1e-4)
        ddp = GradBuckets([dn, pn], bucket_bytes=2 << 20, transport_dtype=transport) if with_ddp else None
        opt.zero_grad()
        hnn.dcdp_forward(dn, pn, b["tgt"], b["ref"], b["K"])[0].backward()
        if ddp is not None:
            ddp.finish()
        for n in (dn, pn):
            n.join_side()
        torch.cuda.synchronize()
        out = (dn.flat_grad.clone(), pn.flat_grad.clone())
        if ddp is not None:
            ddp.detach()
        return out

    plain = grads(None, with_ddp=False)
    f32 = grads(None)
    bf16 = grads(torch.bfloat16)
    for name, p, a, c in zip(("DepthNet", "PoseNet"), plain, f32, bf16):
        assert p.abs().max().item() > 0
        assert torch.equal(a, p), f"{name}: the fp32-transport gradient is not bitwise the gradient without a process group"
        assert torch.equal(c, p.to(torch.bfloat16).float()), f"{name}: the bf16-transport gradient is not the bf16 rounding of it"
        assert not torch.equal(c, p)
    print("RCCL_TRANSPORT_OK", flush=True)
    os._exit(0)


if __name__ == "__main__":
    main()
