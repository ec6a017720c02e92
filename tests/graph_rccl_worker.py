"""Worker for tests/test_graph_gpu.py::test_graphed_step_with_rccl_single_rank: ONE process per gradient-transport dtype.
The hipGraph-captured data-parallel step (RCCL, one rank) against the eager data-parallel step and against the step without any
process group, in deterministic mode: bit for bit.  Its own process: the HIP runtime ends a failed stream capture with abort(), which
must not take the test session down with it, and a process group is process-global state."""
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    transport = {"f32": None, "bf16": torch.bfloat16}[sys.argv[1]]
    from coivo_amd import nn as hnn
    from coivo_amd import synth
    from coivo_amd.ddp import GradBuckets
    from coivo_amd.graph import GraphedTrainStep
    from coivo_amd.optim import FusedAdam
    from oracle import colvo_spec as S
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)

    def setup(seed):
        dn_o, pn_o = S.make_models(seed)
        dn, pn = hnn.DepthNet(compute_dtype=torch.bfloat16), hnn.PoseNet(compute_dtype=torch.bfloat16)
        dn.load_state_dict(dn_o.state_dict())
        pn.load_state_dict(pn_o.state_dict())
        dn.deterministic = pn.deterministic = True
        return dn, pn, FusedAdam([dn, pn], lr=1e-4)

    # default: B=2 64x96; `... f32 64 256 320` = the per-GPU shape of BASELINE configs[4] (64 pairs of 320x256): from 32 pairs on
    # GraphedTrainStep cuts the weight-gradient chain into one-command segments (graph.py), a branch the small shape never reaches
    B, H, W = (int(v) for v in sys.argv[2:5]) if len(sys.argv) >= 5 and sys.argv[2].isdigit() else (2, 64, 96)
    seed = 63
    b = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in synth.make_batch(B, H, W, seed=seed).items()}
    frames = torch.cat([b["tgt"], b["ref"]])
    dn1, pn1, opt1 = setup(seed)
    dn2, pn2, opt2 = setup(seed)
    # (RCCL called natively on the group's communicator, ddp._NativeRccl: this worker runs no torch collective on the group; `... torch`
    #  as the last argument keeps ProcessGroup.allreduce)
    native = sys.argv[-1] != "torch"
    # `... defer` as the last argument: the loss normaliser's exchange off the critical path (GradBuckets(defer_loss_normalisation=True):
    # what bench.py runs data parallel since round 6) -- the two-float all-reduce, colvo_warp_loss_rescale_to in finish() and the
    # optimizer's device-side scale are then all INSIDE the captured step
    defer = sys.argv[-1] == "defer"
    ddp1 = GradBuckets([dn1, pn1], bucket_bytes=4 << 20, transport_dtype=transport, native_collectives=native,
                       **(dict(defer_loss_normalisation=True, optimizer=opt1) if defer else {}))
    ddp2 = GradBuckets([dn2, pn2], bucket_bytes=4 << 20, transport_dtype=transport, native_collectives=native,
                       **(dict(defer_loss_normalisation=True, optimizer=opt2) if defer else {}))
    assert ddp1._defer == defer and ddp2._defer == defer
    assert ddp1.native_collectives == native and ddp2.native_collectives == native
    if not native:
        # the rule (coivo_amd/graph.py _process_group_path): a captured nccl step goes through the native RCCL path; through
        # ProcessGroup.allreduce it is refused -- before anything is captured, and leaving the networks usable -- unless the caller opts in
        refused = GraphedTrainStep(dn2, pn2, opt2, B, H, W, ddp=ddp2)
        try:
            refused.capture()
            raise AssertionError("a captured step through ProcessGroup.allreduce was not refused")
        except RuntimeError as e:
            assert "native_collectives=True" in str(e), e
        # (the refusal comes first in capture(): nothing ran, nothing in the library changed)
        assert refused.graph is None and torch.equal(dn1.flat_param, dn2.flat_param) and torch.equal(pn1.flat_param, pn2.flat_param)
    step = GraphedTrainStep(dn2, pn2, opt2, B, H, W, ddp=ddp2, allow_process_group_capture=not native)
    assert step.capture_group == (1 if B >= 32 else 2)
    eager, graphed = [], []
    for _ in range(3):
        # (photometric_loss's batch reducer is ONE process-wide hook, the last GradBuckets' -- a process has one set of networks; this
        #  worker has two, and with the deferred normalisation the reducer carries state: each step gets its own object's)
        ddp1._install_reducer(True)
        opt1.zero_grad()
        loss = hnn.dcdp_forward(dn1, pn1, b["tgt"], b["ref"], b["K"])[0]
        loss.backward()
        ddp1.finish()
        opt1.step()
        eager.append(loss.item())
        ddp2._install_reducer(True)
        graphed.append(step(frames, b["K"]).item())
    torch.cuda.synchronize()
    assert eager == graphed, (sys.argv[1], eager, graphed)
    assert torch.equal(dn1.flat_param, dn2.flat_param) and torch.equal(pn1.flat_param, pn2.flat_param), sys.argv[1]
    assert graphed[-1] < graphed[0]
    st = step.stats
    assert st["pending_commands"] == 0 and st["side_commands"] == 29, st      # (deterministic nets: no fused iconv1 kernel; the head's MFMA form is 2)
    if transport is None and not defer:
        # ... and the fp32-transport run is bitwise the run WITHOUT any process group (a one-rank all-reduce is the identity; not with
        # the deferred normalisation, which applies the scale in the optimizer instead of the heads' backward kernels: other roundings)
        dn3, pn3, opt3 = setup(seed)
        for _ in range(3):
            opt3.zero_grad()
            hnn.dcdp_forward(dn3, pn3, b["tgt"], b["ref"], b["K"])[0].backward()
            opt3.step()
        torch.cuda.synchronize()
        assert torch.equal(dn1.flat_param, dn3.flat_param) and torch.equal(pn1.flat_param, pn3.flat_param)
    ddp1.detach()
    ddp2.detach()
    torch.cuda.synchronize()
    print(f"GRAPH_RCCL_OK {sys.argv[1]}{' DEFER' if defer else ''} losses {graphed}", flush=True)
    sys.stdout.flush()
    # leave without tearing the communicator / the graph down piece by piece: the process is done
    os._exit(0)


if __name__ == "__main__":
    main()
