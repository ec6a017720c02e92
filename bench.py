#!/usr/bin/env python3
"""bench.py -- training frame-pairs/sec of the ColVO DCDP+LCC step on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A "step" = one full training step on one batch of synthetic frame pairs already resident in HBM:
DepthNet (both frames) + PoseNet forward, fused warp/LCC/SSIM/L1 loss, backward through everything,
gradient all-reduce (N > 1), Adam.  N = 1 runs BASELINE configs[1]: batch 8, 320x256, bf16 conv / fp32 loss.
Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline` (the fused warp/loss op at BASELINE configs[2],
where SURVEY.md section 8d reads the HBM roofline; `roofline_in_step`: the same op inside this run's steps) and `cpu_baseline`.
Protocol (round 4): W warm-up steps, barrier + synchronize, EXACTLY K steps and nothing else, barrier + synchronize; then, outside
the timed region: the same K steps started without draining (`ms_per_step_pipeline_full`), the three call-sequence forms
interleaved A-B-C in blocks (`forms_interleaved`), the fused op bracketed inside 10 extra steps, the op alone at configs[2].
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

if int(os.environ.get("WORLD_SIZE", "1")) > 1 or "--rccl-single" in sys.argv:
    # Data parallel: the main stream, the weight-gradient side stream and RCCL's stream must not share a hardware queue.  With the
    # runtime's default of 4 queues the side stream landed on the main stream's queue once the communicator existed and the
    # backward pass serialised (1.96 ms per step against 1.68 with 8 queues, one rank through the RCCL path).  Read at HIP
    # initialisation, so it has to be in the environment before torch is imported.  A value the user exported is left alone
    # (coivo_amd.streams warns about 3..7); the effective value is reported in the JSON line (`hw_queues`).
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
    # (a captured step, configs[4], holds RCCL collectives: what makes that safe is the native RCCL path -- no ProcessGroupNCCL work
    #  objects, torch's RCCL stream never in a capture -- coivo_amd/graph.py _process_group_path; round 5's
    #  TORCH_NCCL_CUDA_EVENT_CACHE=0 default belonged to a hypothesis that was retired and is gone)

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s spec
LOSS_BYTES_PER_PIXEL = 60.0    # fwd 28 (tgt 12 + ref 12 + depth 4) + bwd 32 (same + d_depth 4)
# What the ONE-PASS kernel must really move: tgt 12 + ref 12 + depth 4 read once, d_depth 4 written = 32 B per pixel.  The §8d
# model counts the inputs twice (a separate backward pass re-reading them), which this build no longer does; `achieved` / `frac`
# keep the §8d definition the contract prescribes, `achieved_real_bytes` / `frac_real_bytes` say how fast bytes actually move.
LOSS_REAL_BYTES_PER_PIXEL = 32.0


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--config", choices=["auto", "1", "2", "3", "4"], default="auto",
                    help="BASELINE.json configs[i]: 1 = 8 pairs/GPU 320x256 bf16 (the N=1 default); 2 = 32 pairs 640x512; 3 = 32 "
                         "pairs/GPU 320x256, fp32 gradient transport (the default for --gpus N > 1: batch 256 on 8 GPUs); 4 = 64 "
                         "pairs/GPU, bf16 gradient transport, hipGraph-captured step (batch 512 on 8 GPUs)")
    ap.add_argument("--batch-per-gpu", type=int, default=None, help="override the configuration's pairs per GPU")
    ap.add_argument("--height", type=int, default=None)
    ap.add_argument("--width", type=int, default=None)
    ap.add_argument("--dtype", choices=["bf16", "f32"], default="bf16")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=20.0, help="budget of the bounded CPU-baseline sample")
    ap.add_argument("--no-roofline-cfg2", action="store_true",
                    help="skip the fused-loss measurement at BASELINE configs[2] (then `roofline` falls back to the in-step figure)")
    ap.add_argument("--no-zero-in-step", action="store_true", help="developer A/B: FusedAdam without zero_grad_in_step (a clearing launch per step)")
    ap.add_argument("--no-side-measurements", action="store_true",
                    help="skip everything outside the K timed steps (interleaved call-sequence forms, in-step fused-op brackets, "
                         "scaling denominator): the process then runs warm-up + K steps only (tests compare final_loss bit for bit)")
    ap.add_argument("--forms-block", type=int, default=10, help="steps per block of the interleaved A-B-C side measurement")
    ap.add_argument("--forms-rounds", type=int, default=3, help="rounds of the interleaved A-B-C side measurement")
    ap.add_argument("--full-loss", action="store_true",
                    help="train on the widened objective (SURVEY.md 8f-1/2: 3-scale photometric + geometric consistency + "
                         "smoothness) instead of BASELINE's plain DCDP+LCC step; reported as such in config.workload")
    ap.add_argument("--bucket-mb", type=int, default=16)
    ap.add_argument("--per-rank-loss", action="store_true",
                    help="developer A/B: data parallel with the mean of the per-rank masked means (rounds 1-4) instead of the spec's "
                         "ONE masked mean over the whole batch (one all-reduce of two floats behind the loss kernel, coivo_amd/ddp.py)")
    ap.add_argument("--blocking-loss-exchange", action="store_true",
                    help="developer A/B: wait for the two-float all-reduce of the loss normaliser between forward and backward (round 5)")
    ap.add_argument("--grad-transport", choices=["f32", "bf16"], default=None, help="override the configuration's transport dtype")
    ap.add_argument("--spec-calls", action="store_true",
                    help="time the step written as the spec's verbatim call sequence (INTEGRATION.md section 1, first snippet: "
                         "depth_net(cat), slicing, photometric_loss on ordinary tensors) instead of the fast path "
                         "(forward_pair_split + gradient handover); without this flag the N=1 line still reports it as "
                         "spec_sequence_ms beside ms_per_step")
    ap.add_argument("--rccl-single", action="store_true",
                    help="test hook: one rank, but through the REAL multi-GPU code path -- init_process_group('nccl', world_size=1), "
                         "GradBuckets, the all-reduces issued from the weight-gradient side stream.  The only way to execute the "
                         "RCCL plumbing on a one-GPU box; the collectives are trivial, the stream interplay is not")
    ap.add_argument("--rehearse-on-one-gpu", action="store_true",
                    help="test hook: all ranks share cuda:0 and talk over gloo (RCCL cannot place two ranks on one device)")
    ap.add_argument("--graph", choices=["auto", "on", "off", "best"], default="auto",
                    help="replay the step from one captured hipGraph (coivo_amd/graph.py); best = time 20 eager steps and, only if the host is their "
                         "limit (enqueue time above 95 %% of the step), capture the graph and replay it (the replay is 2-4 %% behind eager on "
                         "an idle host and immune to a busy one: the eager step needs ~0.9 ms of host time per 1.5 ms step); auto = the "
                         "configuration's setting: on for configs[4], best on one GPU, off on several")
    ap.add_argument("--graph-policy", type=int, choices=[0, 1, 2, 3], default=2,
                    help="how the weight-gradient chain hangs off the main chain in the graph (include/colvo.h "
                         "colvo_set_capture_policy): 0 one branch, 1 one edge per layer, 2 segments of --graph-group commands")
    ap.add_argument("--graph-group", type=int, default=None, help="default: 1 from 32 pairs per GPU on, else 2 (coivo_amd/graph.py)")
    return resolve_config(ap.parse_args())


CONFIGS = {      # BASELINE.json configs[i] -> (pairs per GPU, H, W, gradient transport, hipGraph)
    "1": (8, 256, 320, "f32", "auto"),
    "2": (32, 512, 640, "f32", "auto"),
    "3": (32, 256, 320, "f32", "auto"),
    "4": (64, 256, 320, "bf16", "on"),
}


def resolve_config(args):
    """--config auto: N=1 -> configs[1] exactly as in rounds 1-2, N>1 -> configs[3] (scaling is read at 32
    pairs per GPU); explicit --batch-per-gpu / --height / --width / --grad-transport / --graph win over the configuration."""
    cfg = args.config if args.config != "auto" else ("1" if args.gpus == 1 else "3")
    b, h, w, transport, graph = CONFIGS[cfg]
    args.config_id = cfg
    args.custom_shape = any(v is not None for v in (args.batch_per_gpu, args.height, args.width))
    args.batch_per_gpu = b if args.batch_per_gpu is None else args.batch_per_gpu
    args.height = h if args.height is None else args.height
    args.width = w if args.width is None else args.width
    args.grad_transport = transport if args.grad_transport is None else args.grad_transport
    if args.graph == "auto":
        # one GPU: whichever form is faster on this host; several: eager unless the configuration names the graph (the replayed
        # RCCL path has run with one rank only)
        args.graph = graph if (graph == "on" or args.gpus > 1 or args.rccl_single) else "best"
    return args


def host_cores() -> int:
    """CPU threads this process may really use: the cgroup quota when there is one (the GPU box exposes all
    host cores in the affinity mask but grants a 16-CPU share per GPU), else the affinity mask."""
    try:
        n = len(os.sched_getaffinity(0))
    except Exception:
        n = os.cpu_count() or 1
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            quota, period = f.read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    if n > 64:          # no quota visible: do not oversubscribe a shared host
        n = 16
    return n


def cpu_baseline(B, H, W, budget_s):
    """The oracle's full train step (fp32, torch CPU) on this box's host cores: a bounded sample."""
    from coivo_amd import synth
    from oracle import colvo_spec as S
    cores = host_cores()
    torch.set_num_threads(cores)
    dn, pn = S.make_models(0)
    opt = torch.optim.Adam(list(dn.parameters()) + list(pn.parameters()), **S.ADAM_KW)
    b = synth.make_batch(B, H, W, seed=1234)
    t0 = time.perf_counter()
    for _ in range(3):                                             # >= 3 warm-ups: allocations, thread pool
        S.train_step(dn, pn, opt, b["tgt"], b["ref"], b["K"])
    warm = time.perf_counter() - t0
    times = []
    t_start = time.perf_counter()
    while len(times) < 10 and (time.perf_counter() - t_start) < max(budget_s - warm, 0.0) or not times:
        t1 = time.perf_counter()
        S.train_step(dn, pn, opt, b["tgt"], b["ref"], b["K"])
        times.append(time.perf_counter() - t1)
    times.sort()
    med = times[len(times) // 2]
    return {"value": B / med, "unit": "frame-pairs/s", "cores": cores, "kind": "port",
            "sample": f"oracle (pure-torch fp32) full train step, B={B} {W}x{H}, 3 warm-ups + {len(times)} timed steps, median"}


def depth_l1_vs_oracle(dev, cdt, B, H, W):
    """BASELINE.json's second metric: mean |depth_hip - depth_oracle| on identical inputs and identical weights (the
    bench's compute dtype; fp32 mode is pinned to 1e-4 by tests/test_nets_gpu.py)."""
    from coivo_amd import nn as hnn
    from coivo_amd import synth
    from oracle import colvo_spec as S
    dn_o, _ = S.make_models(0)
    dn = hnn.DepthNet(compute_dtype=cdt, device=dev)
    dn.load_state_dict(dn_o.state_dict())
    b = synth.make_batch(B, H, W, seed=1234)
    x = torch.cat([b["tgt"], b["ref"]])
    with torch.no_grad():
        do = dn_o(x)
        dh = dn(x.to(dev)).cpu()
    err = (dh - do).abs()
    return {"mean_abs": err.mean().item(), "max_abs": err.max().item(), "mean_rel": (err.mean() / do.abs().mean()).item(),
            "images": int(x.shape[0]), "weights": "spec init, seed 0", "dtype": "bf16" if cdt == torch.bfloat16 else "f32"}


def conv_flops_per_step(B, H, W):
    """Algorithmic conv flops of one training step: forward + input gradient + weight gradient."""
    from coivo_amd import nn as hnn
    f, h, w, cin = 0, H, W, 3
    for c in hnn.ENC_CH:                               # DepthNet runs on 2B images
        h, w = h // 2, w // 2
        f += 2 * 9 * h * w * (cin * c + c * c); cin = c
    for i in range(5, 0, -1):
        d = hnn.DEC_CH[i - 1]; h, w = h * 2, w * 2
        skip = hnn.ENC_CH[i - 2] if i >= 2 else 0
        f += 2 * 9 * h * w * (cin * d + (d + skip) * d); cin = d
    f += 2 * 9 * h * w * cin
    f *= 2 * B
    g, h, w, cin = 0, H, W, 8
    for c in hnn.POSE_CH:
        h, w = (h - 1) // 2 + 1, (w - 1) // 2 + 1
        g += 2 * 9 * h * w * cin * c; cin = c
    return 3 * (f + g * B)


def pmc_traffic(workload):
    """HBM bytes of the fused loss (all its kernels, per step) from the committed PMC summary (profiles/rN_traffic.json), or None."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_traffic.json")))
    if not files:
        return None
    try:
        ks = json.load(open(files[-1]))["kernels"]
        sel = [k for k in ks if k["workload"] == workload]
        return sum(k["hbm_bytes"] for k in sel) if sel else None     # r1: fwd + bwd kernels; r2 on: the one-pass kernel
    except Exception:
        return None


def roofline_cfg2(dev):
    """Fused warp/loss fwd+bwd alone at BASELINE configs[2] (B=32, 640x512): where SURVEY.md §8d reads the HBM roofline."""
    from coivo_amd import functional as Fh
    from coivo_amd import synth
    B, H, W = 32, 512, 640
    b = synth.make_batch(4, H, W, seed=77, device=dev)
    rep = lambda t: t.repeat(B // 4, *([1] * (t.dim() - 1))).contiguous()
    tgt, ref, K = rep(b["tgt"]), rep(b["ref"]), rep(b["K"])
    depth = rep(b["gt_depth"]).requires_grad_(True)
    pose = rep(b["gt_pose"]).requires_grad_(True)
    a = rep(b["gt_a"]).requires_grad_(True)
    bb = rep(b["gt_b"]).requires_grad_(True)
    # the training form of the op (nn.DepthNet.forward_pair_split): the depth gradient leaves the op unnormalised together
    # with two device scalars, and its consumer -- the depth head's backward kernel -- multiplies while it reads it
    hand, hand_p = Fh.GradHandover(), Fh.GradHandover()
    depth._colvo_handover = hand
    pose._colvo_handover = a._colvo_handover = bb._colvo_handover = hand_p
    def one():
        loss = Fh.photometric_loss(tgt, ref, depth, pose, K, a, bb)
        g_raw, gp, ga, gb = torch.autograd.grad(loss, [depth, pose, a, bb])
        hand.take((g_raw,))
        hand_p.take((gp, ga, gb))

    Fh.enable_timing(True)
    for it in range(25):
        one()
    torch.cuda.synchronize()
    ev = Fh.timing_events()
    tf = sorted(e0.elapsed_time(e1) for e0, e1 in ev["fwd"][5:])
    tb = sorted(e0.elapsed_time(e1) for e0, e1 in ev["bwd"][5:])
    Fh.enable_timing(False)
    f_single_ms, b_ms = tf[len(tf) // 2], tb[len(tb) // 2]
    # The op's duration proper: NB C-ABI calls back to back between ONE pair of events on the launch stream (the host enqueues a
    # call in microseconds, the GPU needs ~0.19 ms: the queue never runs dry), divided by NB.  A bracket around every single call adds the two
    # event records and the launch gap behind an idle queue to each sample (~12 us here: fwd_us_single_bracket) -- rocprofv3's
    # average for the march kernel + finalize (profiles/r3_bench_kernel_stats.csv, r3_traffic.json) agrees with THIS figure.
    NB = 20
    from coivo_amd import _lib
    lib = _lib.load()
    f32 = dict(device=dev, dtype=torch.float32)
    ws = torch.empty(lib.colvo_warp_loss_workspace_floats(B, H, W), **f32)
    state, d_raw = torch.empty(4, **f32), torch.empty(B, 1, H, W, **f32)
    gpart, gunit = torch.empty(B * 14, **f32), torch.empty(B * 8, **f32)
    dd, pp, aa, b2 = depth.detach(), pose.detach(), a.detach(), bb.detach()
    sp = torch.cuda.current_stream().cuda_stream

    def call():         # exactly the C-ABI call functional._WarpLoss.forward makes (the one the single brackets sit around)
        _lib.check(lib.colvo_warp_loss_fused(_lib.ptr(tgt), _lib.ptr(ref), _lib.ptr(dd), _lib.ptr(pp), _lib.ptr(K), _lib.ptr(aa),
                                             _lib.ptr(b2), B, H, W, Fh.SSIM_WEIGHT, _lib.ptr(ws), _lib.ptr(state), _lib.ptr(d_raw),
                                             _lib.ptr(gpart), _lib.ptr(gunit), sp), "colvo_warp_loss_fused")
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = []
    for _ in range(7):          # the VALU-bound kernel follows the clocks: repetitions on one box spread by +-4 %, hence seven
        torch.cuda.synchronize()
        e0.record()
        for it in range(NB):
            call()
        e1.record()
        torch.cuda.synchronize()
        reps.append(e0.elapsed_time(e1) / NB)
    f_ms = sorted(reps)[len(reps) // 2]
    px = B * H * W
    # the backward call launches NOTHING (gradient handover): the op's duration is the forward call's; the bracket around the
    # empty backward measures the two event records themselves (~4.5 us, reported as bwd_us) and is not a kernel duration
    ach = LOSS_BYTES_PER_PIXEL * px / (f_ms * 1e-3) / 1e9
    return {"kernel": "k_warp_loss_bwd_march<fused> + k_warp_loss_fused_finalize (loss and all gradients in one pass)",
            "workload": f"B={B} {W}x{H} fp32, 1 warp direction", "bwd_launches": 0,
            "bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
            "traffic": pmc_traffic("B=32 640x512 (configs[2])"),
            "fwd_us": f_ms * 1e3, "fwd_us_min": min(reps) * 1e3, "fwd_us_max": max(reps) * 1e3,
            "fwd_us_single_bracket": f_single_ms * 1e3, "bwd_us": b_ms * 1e3,
            "algorithmic_bytes": LOSS_BYTES_PER_PIXEL * px, "real_bytes": LOSS_REAL_BYTES_PER_PIXEL * px,
            "achieved_real_bytes": LOSS_REAL_BYTES_PER_PIXEL * px / (f_ms * 1e-3) / 1e9,
            "frac_real_bytes": LOSS_REAL_BYTES_PER_PIXEL * px / (f_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
            "timing": "fwd_us: 20 calls of the op back to back between ONE pair of hip events on the launch stream, / 20, median of 7 "
                      "repetitions (fwd_us_min / fwd_us_max: their range; forward call = one-pass loss + unnormalised gradients + finalize; the backward call launches "
                      "nothing -- all four gradients are handed to their consumers, the depth / pose head backward kernels, "
                      "unnormalised with two device scalars, as in the training step).  fwd_us_single_bracket: the round-1/2 "
                      "protocol, an event pair around every single call (adds the two event records and the launch gap behind an "
                      "idle queue); bwd_us: such a pair around the empty backward call = the cost of the bracket itself"}


def workload_name(args, B, H, W):
    what = {"1": "BASELINE configs[1]", "2": "BASELINE configs[2] shape", "3": "BASELINE configs[3] (per-GPU share of batch 256 on 8 GPUs)",
            "4": "BASELINE configs[4] (per-GPU share of batch 512 on 8 GPUs)"}[args.config_id]
    if args.custom_shape:
        what = "custom shape"
    return (("WIDENED OBJECTIVE (not BASELINE's metric): 3-scale photometric + geometric consistency + smoothness; "
             if args.full_loss else "") +
            f"{what}: batch={B}/GPU {W}x{H} full DCDP+LCC train step (DepthNet x2 frames + PoseNet fwd/bwd, fused "
            f"warp/LCC/SSIM/L1 loss fwd/bwd, Adam), {args.dtype} conv / fp32 loss")


def plan_distributed(args, env):
    """Everything the multi-process launch derives from the command line and the torchrun environment, as plain data
    (tests/test_bench_cpu.py walks the RCCL branch up to -- not including -- init_process_group without a GPU)."""
    rank = int(env.get("RANK", "0"))
    local_rank = int(env.get("LOCAL_RANK", "0"))
    world = int(env.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if args.gpus > 1 and world == 1:
        raise SystemExit("for --gpus N > 1 launch with: python -m torch.distributed.run --nproc-per-node N bench.py ...")
    if args.rehearse_on_one_gpu:
        local_rank = 0
    plan = {"rank": rank, "local_rank": local_rank, "world": world, "device": ("cuda", local_rank), "backend": None,
            "init_kwargs": None, "master_addr": env.get("MASTER_ADDR", "127.0.0.1"), "seed": 1234 + rank,
            "global_batch": world * args.batch_per_gpu,
            "grad_transport": (args.grad_transport if world > 1 else None)}
    if world > 1:
        plan["backend"] = "gloo" if args.rehearse_on_one_gpu else "nccl"      # "nccl" IS RCCL on ROCm
        plan["init_kwargs"] = {"rank": rank, "world_size": world}
        if plan["backend"] == "nccl":
            plan["init_kwargs"]["device_id"] = plan["device"]
    return plan


def main():
    args = parse()
    plan = plan_distributed(args, os.environ)
    rank, local_rank, world = plan["rank"], plan["local_rank"], plan["world"]
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (torch.cuda is not available); there is no CPU fallback")
    torch.cuda.set_device(local_rank)
    dev = torch.device(*plan["device"])

    import contextlib
    import torch.distributed as dist

    @contextlib.contextmanager
    def stdout_to_stderr():
        """RCCL prints a version banner on STDOUT when its first communicator is created; stdout carries exactly one JSON line
        (the driver parses it), so the file descriptor points at stderr while the communicator comes up."""
        sys.stdout.flush()
        saved = os.dup(1)
        os.dup2(2, 1)
        try:
            yield
        finally:
            sys.stdout.flush()
            os.dup2(saved, 1)
            os.close(saved)

    with stdout_to_stderr():
        if args.rccl_single and world == 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29517")
            dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
        if world > 1:
            os.environ.setdefault("MASTER_ADDR", plan["master_addr"])
            kw = dict(plan["init_kwargs"])
            if "device_id" in kw:
                kw["device_id"] = dev
            dist.init_process_group(plan["backend"], **kw)
        host_pg = None
        if dist.is_initialized() and dist.get_backend() == "nccl":
            # RCCL runs through its own C ABI on the group's communicator (coivo_amd.ddp._NativeRccl): torch must then keep its own
            # collectives off that communicator -- they would run on torch's internal stream, a fifth hardware queue beside the
            # step's four (2.7 x slower steps).  The ranks' host-side synchronisation (barriers around the timed region, the max of
            # the elapsed times) goes over a gloo side group; where that cannot be made the ProcessGroup path stays as it was.
            if os.environ.get("COLVO_DDP_TORCH_COLLECTIVES", "0") in ("", "0"):
                try:
                    if os.environ.get("MASTER_ADDR", "127.0.0.1") in ("127.0.0.1", "localhost"):
                        os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")      # one node: do not depend on the host name resolving
                    host_pg = dist.new_group(backend="gloo")
                except Exception as e:          # noqa: BLE001
                    print(f"[bench] no gloo side group ({type(e).__name__}: {e}): RCCL through ProcessGroup.allreduce", file=sys.stderr)
        if dist.is_initialized() and host_pg is None:
            # the first collective creates the communicator (and prints the banner): do it here, on a throw-away tensor
            t0_ = torch.zeros(1, device=dev if plan["backend"] != "gloo" or args.rccl_single else "cpu")
            dist.all_reduce(t0_)
            if t0_.is_cuda:
                torch.cuda.synchronize()

    def rank_barrier():
        """Every rank has reached this point (host side over gloo when RCCL is driven natively, else the group's own barrier)."""
        if world > 1:
            if host_pg is not None:
                dist.barrier(group=host_pg)
            else:
                dist.barrier()

    from coivo_amd import build as _colvo_build      # fresh checkout / edited kernels: (re)build in-tree, rank 0 first
    if local_rank == 0:
        _colvo_build.ensure()
    rank_barrier()
    from coivo_amd import functional as Fh
    from coivo_amd import nn as hnn
    from coivo_amd import synth
    from coivo_amd.ddp import GradBuckets
    from coivo_amd.optim import FusedAdam

    B, H, W = args.batch_per_gpu, args.height, args.width
    cdt = torch.bfloat16 if args.dtype == "bf16" else torch.float32
    dn, pn = hnn.DepthNet(compute_dtype=cdt, device=dev), hnn.PoseNet(compute_dtype=cdt, device=dev)
    # random-init weights of the spec'd architecture, identical on every rank (same seed)
    g = torch.Generator().manual_seed(0)
    with torch.no_grad():
        for net in (dn, pn):
            for name, p in net.named_parameters():
                if name.endswith("weight"):
                    fan_in = p.shape[1] * p.shape[2] * p.shape[3]
                    p.copy_((torch.randn(p.shape, generator=g) * (2.0 / fan_in) ** 0.5).to(dev))
    # (step() clears the gradients it has just read: the zero_grad() that opens each step then costs no launch -- optim.py)
    opt = FusedAdam([dn, pn], lr=1e-4, zero_grad_in_step=not args.no_zero_in_step)
    ddp = None
    if world > 1 or args.rccl_single:
        ddp = GradBuckets([dn, pn], bucket_bytes=args.bucket_mb << 20,
                          transport_dtype=torch.bfloat16 if args.grad_transport == "bf16" else None,
                          exact_batch_loss=not args.per_rank_loss, native_collectives=host_pg is not None,
                          # the two-float loss exchange off the critical path (round 6; the plain objective is ONE photometric_loss
                          # call and the only source of gradients: what the option asks of its caller)
                          defer_loss_normalisation=not (args.per_rank_loss or args.full_loss or args.blocking_loss_exchange),
                          optimizer=opt)
        opt.grad_scale = ddp.grad_scale
    batch = synth.make_batch(B, H, W, seed=1234 + rank, device=dev)
    frames = torch.cat([batch["tgt"], batch["ref"]], dim=0)      # one resident buffer: target frames, then reference
    tgt, ref, K = frames[:B], frames[B:], batch["K"]

    one = torch.ones((), device=dev)      # dL/dloss, persistent (no ones_like() fill kernel per step)
    graphed = None
    graph_error = graph_trial = None

    def capture_graph():
        nonlocal graphed, graph_error
        from coivo_amd.graph import GraphedTrainStep
        try:
            # (with the ProcessGroup.allreduce fallback -- COLVO_DDP_TORCH_COLLECTIVES=1 or a torch without _comm_ptr() -- the capture is
            #  refused by GraphedTrainStep and the run falls back to eager launches, saying so in the JSON line)
            graphed = GraphedTrainStep(dn, pn, opt, B, H, W, ddp=ddp, capture_policy=args.graph_policy,
                                       capture_group=args.graph_group, full_loss=args.full_loss)
            graphed.frames.copy_(frames)
            graphed.K.copy_(K)
            graphed.capture()
        except Exception as e:                       # noqa: BLE001 -- fall back to eager launches, say so in the JSON
            if args.graph == "on":
                raise
            graphed = None
            graph_error = f"{type(e).__name__}: {e}"[:200]
            torch.cuda.synchronize()

    if args.graph == "on" and not args.spec_calls:
        capture_graph()

    def fast_step(full=args.full_loss):
        opt.zero_grad()
        if full:            # the widened objective: two native calls, no torch kernels
            loss = hnn.dcdp_forward(dn, pn, None, None, K, full_loss=True, frames=frames)[0]
        else:
            d_t, d_r, d_l = dn.forward_pair_split(frames)     # d_l: depth_t again, the loss's own gradient path (nn.py)
            pose, a, b = pn(tgt, ref, d_t, d_r)
            loss = Fh.photometric_loss(tgt, ref, d_l, pose, K, a, b)
        loss.backward(gradient=one)
        if ddp is not None and ddp.attached:
            ddp.finish()
        opt.step()
        return loss

    def spec_step():
        """The spec's call sequence verbatim (first snippet; oracle/colvo_spec.py train_step): one
        batched DepthNet forward, slicing, photometric_loss on ordinary tensors, plain loss.backward()."""
        opt.zero_grad()
        d = dn(torch.cat([tgt, ref]))
        d_t, d_r = d[:B], d[B:]
        pose, a, b = pn(tgt, ref, d_t, d_r)
        loss = Fh.photometric_loss(tgt, ref, d_t, pose, K, a, b)
        loss.backward()
        if ddp is not None and ddp.attached:
            ddp.finish()
        opt.step()
        return loss

    eager_step = spec_step if args.spec_calls else fast_step

    def step(timed: bool):
        if graphed is not None and not timed:
            return graphed()
        return eager_step()

    def timed_run(fn, n, warm=3):
        """n calls of fn after `warm` untimed ones, wall clock between two device synchronisations -> ms per call."""
        for _ in range(warm):
            fn()
        torch.cuda.synchronize()
        t_ = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t_) / n * 1e3

    def barrier():
        # (device drained first: with the host-side barrier nothing of it is queued behind the step)
        torch.cuda.synchronize()
        rank_barrier()
        torch.cuda.synchronize()

    if args.graph == "best" and not args.spec_calls:
        # Eager launches need the host to stay ahead of the GPU (~0.9 ms of enqueue work per 1.5 ms step at configs[1]); on a busy
        # host they do not, and the replayed graph -- level with eager otherwise, within 2 % -- is immune.  Decide from 20 eager
        # steps: host-bound (enqueue time above 95 % of the step: the GPU waits for the host) -> capture and replay -- not earlier:
        # at 82 % the eager step still ran at its 1.48 ms and the replay chosen by an 80 % rule cost 6 %.  The decision must come BEFORE a
        # capture: the streams a capture leaves behind push eager steps of the same process over the hardware-queue cliff
        # (measured 4.6 ms per eager step after a capture).
        # (round 5: the trial runs with the cyclic collector off, like the timed region -- a collection inside the 20 steps read as
        #  1.27 ms of host time per step and sent a default run to the replay, 22 % slower -- and the replay is kept only where it
        #  MEASURES faster than the eager steps: GraphedTrainStep.close() returns the process to its pre-capture state)
        import gc
        gc.collect()
        gc.disable()
        try:
            for _ in range(3):
                eager_step()
            torch.cuda.synchronize()
            t_ = time.perf_counter()
            for _ in range(20):
                eager_step()
            t_host = (time.perf_counter() - t_) / 20 * 1e3
            torch.cuda.synchronize()
            t_eager = (time.perf_counter() - t_) / 20 * 1e3
            graph_trial = {"eager_ms": t_eager, "eager_host_enqueue_ms": t_host, "steps": 20, "chosen": "eager"}
            if t_host > float(os.environ.get("COLVO_BENCH_GRAPH_THRESHOLD", "0.95")) * t_eager:     # (developer probe: 0 forces the replay)
                capture_graph()
                if graphed is not None:
                    t_replay = timed_run(graphed, 20)
                    graph_trial.update(replay_ms=t_replay, chosen="replay")
                    if t_replay >= 0.98 * t_eager and os.environ.get("COLVO_BENCH_GRAPH_THRESHOLD") is None:
                        graphed.close()
                        graphed = None
                        graph_trial.update(chosen="eager (the replay measured no faster)")
        finally:
            gc.enable()
    use_graph = graphed is not None
    first_loss = None
    # a cyclic-GC pass of the interpreter inside the K steps is a multi-millisecond host stall that has nothing to do with the
    # path: collect now and keep the collector off until the timed region is over.  BEFORE the warm-up, not between warm-up and timed
    # region as rounds 3-5 did: a full collection walks every object of the process and leaves the interpreter's working set cold --
    # the first step after it took 1.3-1.6 ms of host time instead of 0.9 and 1.65-1.9 ms on the GPU instead of 1.3 (2.0-2.6 ms in this
    # script's larger heap), which a 20-step run from a drained GPU sees as +3 % per step (tools/first_step_probe.py)
    import gc
    step_ev = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    gc.collect()
    gc.disable()
    for i in range(args.warmup):
        l_ = step(False)
        if i == 0:
            first_loss = l_                       # read after the timed region (no host synchronisation inside the warm-up)
    barrier()
    t0 = time.perf_counter()
    # The headline: EXACTLY K steps between barrier + synchronize on both sides (the task's contract), nothing else inside -- the
    # hip-event brackets around the fused op that rounds 1-3 kept in these steps (two event records per call, ~10 us per step) are
    # gone; the op is timed in a side measurement below.  The region starts on a DRAINED GPU (the contract's synchronize): the
    # first step's kernels wait for the host's first enqueues (~0.5 ms of a 20-step run); `ms_per_step_pipeline_full` below is the
    # same loop started without draining.
    # developer probes (tools/README.md): COLVO_BENCH_HOST_SPIN_US burns host time in every step (does the step time move? then
    # the host is the limit), COLVO_BENCH_HOST_LEAD prints how far the host ran ahead of the GPU (host enqueue time per step)
    spin_us = float(os.environ.get("COLVO_BENCH_HOST_SPIN_US", "0"))
    host_t = []
    for i in range(args.steps):
        step_ev[i].record()
        h0 = time.perf_counter()
        loss = step(not use_graph)
        if spin_us > 0:
            t_end = time.perf_counter() + spin_us * 1e-6
            while time.perf_counter() < t_end:
                pass
        host_t.append(time.perf_counter() - h0)
    step_ev[args.steps].record()
    host_done = time.perf_counter() - t0
    barrier()
    elapsed = time.perf_counter() - t0
    gc.enable()
    ev_raw = [step_ev[i].elapsed_time(step_ev[i + 1]) for i in range(args.steps)]
    if os.environ.get("COLVO_BENCH_HOST_LEAD") and rank == 0:
        hs = sorted(host_t)
        print(f"host: enqueue of {args.steps} steps took {host_done * 1e3:.2f} ms of the {elapsed * 1e3:.2f} ms until the GPU "
              f"finished; per step median {hs[len(hs) // 2] * 1e3:.3f} ms, max {hs[-1] * 1e3:.3f} ms", file=sys.stderr)
    if os.environ.get("COLVO_BENCH_DUMP_STEPS") and rank == 0:
        print("step ms:", " ".join(f"{v:.2f}" for v in ev_raw), file=sys.stderr)
    ev_ms = sorted(ev_raw)
    if world > 1:
        t = torch.tensor([elapsed], device=dev if host_pg is None else "cpu", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=host_pg)
        elapsed = t.item()
    final_loss = loss.item()
    first_loss = first_loss.item() if first_loss is not None else final_loss

    def event_medians(plan):
        """plan: [(tag, fn, n)] run back to back WITHOUT draining in between, one hip event per step boundary -> {tag: sorted ms}."""
        evs, tags = [torch.cuda.Event(enable_timing=True)], []
        evs[0].record()
        for tag, fn, n in plan:
            for _ in range(n):
                fn()
                e_ = torch.cuda.Event(enable_timing=True)
                e_.record()
                evs.append(e_)
                tags.append(tag)
        torch.cuda.synchronize()
        out_ = {}
        for i_, tag in enumerate(tags):
            out_.setdefault(tag, []).append(evs[i_].elapsed_time(evs[i_ + 1]))
        return {k: sorted(v) for k, v in out_.items()}

    med = lambda v: v[len(v) // 2]
    main_fn = graphed if use_graph else eager_step
    pipeline_full = None
    if not args.no_side_measurements:
        # the same K steps behind 3 lead-in steps, the start event recorded WITHOUT draining the GPU in front of it (VERDICT r3
        # item 5): what a training loop sees per step once it runs
        gc.collect(); gc.disable()
        m_ = event_medians([("lead", main_fn, 3), ("timed", main_fn, args.steps)])
        gc.enable()
        pipeline_full = sum(m_["timed"]) / len(m_["timed"])

    # ---- side measurements, outside the timed region (every rank runs them: no rank idles at a collective) ----
    side = {}
    if ddp is not None and not use_graph and not args.no_side_measurements:
        # scaling = fps(N GPUs, b pairs each) / fps(1 GPU, b pairs).  The denominator, measured here on
        # this rank's GPU: the same step at the same per-GPU batch with the gradient exchange detached.
        ddp.pause()          # hooks off, but the communicator's hardware queue stays claimed (coivo_amd/streams.py)
        scale_was, opt.grad_scale = opt.grad_scale, 1.0
        ms1 = timed_run(fast_step, min(args.steps, 20))
        side["single_gpu_same_batch"] = {"pairs_per_gpu": B, "ms_per_step": ms1, "value": B / (ms1 * 1e-3),
                                         "what": "the same step on ONE GPU at the same per-GPU batch without the gradient "
                                                 "exchange (rank 0's GPU, 20 steps): the denominator of the scaling ratio"}
        opt.grad_scale = scale_was
        ddp.resume()
        # where in the step the bucket collectives go out: one extra traced step behind two lead-in steps
        for _ in range(2):
            fast_step(False)
        ddp.trace_buckets(True)
        e_start, e_bwd, e_done = (torch.cuda.Event(enable_timing=True) for _ in range(3))
        e_start.record()
        opt.zero_grad()
        d_t_, d_r_, d_l_ = dn.forward_pair_split(frames)
        pose_, a_, b_ = pn(tgt, ref, d_t_, d_r_)
        loss_ = Fh.photometric_loss(tgt, ref, d_l_, pose_, K, a_, b_)
        loss_.backward(gradient=one)
        e_bwd.record()
        ddp.finish()
        opt.step()
        e_done.record()
        torch.cuda.synchronize()
        side["bucket_trace"] = {"buckets": ddp.bucket_trace(e_start), "backward_done_ms": round(e_start.elapsed_time(e_bwd), 4),
                                "step_done_ms": round(e_start.elapsed_time(e_done), 4),
                                "what": "one traced step: hip-event offset (ms after the step's first launch, on the issuing stream) of "
                                        "every gradient bucket's all-reduce, in launch order; backward_done_ms = the main stream behind "
                                        "the backward pass, step_done_ms = behind finish() + the optimizer.  Buckets issued well before "
                                        "backward_done_ms overlap the rest of the backward pass; the time between backward_done_ms and "
                                        "step_done_ms minus the optimizer's (~0.07 ms) is the exposed tail of the exchange"}
        ddp.trace_buckets(False)
    if world == 1 and not use_graph and not args.no_side_measurements and ddp is None:
        # The three forms of the step -- fast path (forward_pair_split + gradient handover), the spec's verbatim call sequence, the
        # widened objective -- INTERLEAVED A-B-C-A-B-C in blocks after a common warm-up, hip-event medians per form: their order is
        # then measurable inside one short driver run (measured one after the other, 20 steps each, the record
        # showed them in the opposite order of DESIGN.md's claim).
        forms = [("fast", lambda: fast_step(False)), ("spec", spec_step), ("full", lambda: fast_step(True))]
        for _, fn in forms:
            for _ in range(3):
                fn()
        gc.collect(); gc.disable()
        # (the first step of a block runs behind another form's tail: it gets a tag of its own and stays out of the medians)
        plan = [(t_, fn, n_) for _ in range(args.forms_rounds) for tag, fn in forms
                for t_, n_ in ((tag + "_first", 1), (tag, max(1, args.forms_block - 1)))]
        m_ = event_medians(plan)
        gc.enable()
        side["forms_interleaved"] = {
            "block": args.forms_block, "rounds": args.forms_rounds,
            "fast_path_ms": med(m_["fast"]), "spec_sequence_ms": med(m_["spec"]), "full_objective_ms": med(m_["full"]),
            "what": "hip-event medians per step of the three forms run A-B-C-A-B-C in blocks without draining in between: fast path "
                    "(the timed form unless --spec-calls / --full-loss), the spec's call sequence (depth_net(cat), slices, "
                    "photometric_loss on ordinary tensors, plain backward), the widened objective (3-scale photometric + geometric "
                    "consistency + smoothness)"}
        side["spec_sequence_ms"] = side["forms_interleaved"]["spec_sequence_ms"]
        side["full_objective"] = {"ms_per_step": side["forms_interleaved"]["full_objective_ms"],
                                  "value": B / (side["forms_interleaved"]["full_objective_ms"] * 1e-3),
                                  "what": "the same step with the widened objective (multi-scale photometric + geometric "
                                          "consistency + smoothness, SURVEY.md section 8f-1/8f-2; bench.py --full-loss times it "
                                          "as the main measurement); from forms_interleaved"}
    if rank == 0 and not args.no_side_measurements:
        # the fused op inside the step: hip-event brackets around its C-ABI calls in 10 EXTRA eager steps (never in the headline's)
        Fh.enable_timing(True)
        for _ in range(10):
            eager_step()
        torch.cuda.synchronize()
    elif not args.no_side_measurements:
        for _ in range(10):
            eager_step()
    rank_barrier()

    if rank == 0:
        ms = elapsed / args.steps * 1e3
        value = world * B * args.steps / elapsed
        roof_in_step = None
        if not args.no_side_measurements:
            ev = Fh.timing_events()
            f_us = [e0.elapsed_time(e1) for e0, e1 in ev["fwd"]]
            b_us = [e0.elapsed_time(e1) for e0, e1 in ev["bwd"]]
            Fh.enable_timing(False)
            # --full-loss: ONE forward call per step runs the whole widened objective (smoothness + pyramid, the one-pass kernel over the
            # three levels with the geometric term inside level 0, finalize) and one backward call the combine kernel; the byte model
            # is the photometric one, sum_s 60 * px / 4^s -- the other terms' bytes are not credited
            f_ms, b_ms = sum(f_us) / len(f_us), sum(b_us) / len(b_us)
            px = B * H * W * (1.0 + 0.25 + 0.0625 if args.full_loss else 1.0)
            # plain step: the backward call launches nothing (gradient handover), so the op's duration is the forward call's;
            # --full-loss: the general path's scaling kernels run in the backward calls and count
            ach = LOSS_BYTES_PER_PIXEL * px / ((f_ms + (b_ms if args.full_loss else 0.0)) * 1e-3) / 1e9
            roof_in_step = {
                "kernel": ("k_full_prepare + k_warp_loss_march_levels<geo> + k_full_finalize, backward k_full_combine (the widened "
                           "objective: every term's value and gradients)" if args.full_loss else
                           "k_warp_loss_bwd_march<fused> + k_warp_loss_fused_finalize (project/sample/LCC/SSIM/L1: loss and all "
                           "gradients in one pass)"),
                "workload": f"B={B} {W}x{H} fp32, inside the training step",
                "bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                # the committed PMC summary holds the two BASELINE shapes; any other invocation (or --full-loss) has none
                "traffic": (None if args.full_loss else
                            pmc_traffic("B=8 320x256 (configs[1])") if (B, H, W) == (8, 256, 320) else
                            pmc_traffic("B=32 640x512 (configs[2])") if (B, H, W) == (32, 512, 640) else None),
                "algorithmic_bytes": LOSS_BYTES_PER_PIXEL * px,
                "real_bytes": LOSS_REAL_BYTES_PER_PIXEL * px,
                "achieved_real_bytes": ach * LOSS_REAL_BYTES_PER_PIXEL / LOSS_BYTES_PER_PIXEL,
                "frac_real_bytes": ach * LOSS_REAL_BYTES_PER_PIXEL / LOSS_BYTES_PER_PIXEL / HBM_PEAK_GBS,
                "fwd_us": f_ms * 1e3, "bwd_us": b_ms * 1e3,
                "timing": "hip events on the launch stream directly around the fused op's C-ABI calls in 10 EXTRA eager steps after the "
                          "timed region (never inside the headline's K steps); forward call = one-pass loss + unnormalised gradients + "
                          "finalize; the backward call launches nothing -- the gradients are normalised by the depth / pose head "
                          "backward kernels -- so bwd_us is the cost of two event records and is not part of `achieved`; mean over "
                          "steps; latency-dominated at this size (SURVEY.md section 8d: the roofline is read at configs[2] = `roofline`)"}
        from coivo_amd import streams as _streams
        out = {"metric": "training frame-pairs/sec at 320x256", "value": value, "unit": "frame-pairs/s",
               "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms,
               "ms_per_step_hipevent_median": ev_ms[len(ev_ms) // 2], "ms_per_step_hipevent_max": ev_ms[-1],
               "ms_per_step_pipeline_full": pipeline_full,
               "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
               "dtype": "bf16" if args.dtype == "bf16" else "f32", "data": "synthetic",
               # (how the timed steps were launched belongs to the workload's name: configs[4] names the hipGraph-captured step, and on
               #  these boxes the replay is the SLOWER form -- DESIGN.md section 7.6 -- so the line says which one it timed)
               "config": {"workload": workload_name(args, B, H, W) + (
                              "; launched as a replayed hipGraph" + (" (as BASELINE configs[4] names it; eager launches measure faster on one GPU)"
                                                                   if args.config_id == "4" else "")
                              if use_graph else "; eager launches"),
                          "baseline_config": args.config_id if not args.custom_shape else None,
                          "global_batch": world * B, "height": H, "width": W,
                          "parallelism": f"dp{world}" + (" (one rank through the RCCL path)" if args.rccl_single else ""),
                          "grad_transport": args.grad_transport if (world > 1 or args.rccl_single) else None,
                          "collectives": (None if ddp is None else "ncclAllReduce on the group's communicator, called natively"
                                          if ddp.native_collectives else "ProcessGroup.allreduce"),
                          # which RCCL, which communicator, and what RCCL itself reports about it (ranks, this rank) -- or why the
                          # native path asked for is not the one in use
                          "native_rccl": (None if ddp is None else ddp._native.describe() if ddp.native_collectives
                                          else {"fallback": ddp.native_fallback}),
                          "batch_loss": (None if ddp is None else "one masked mean over the global batch (valid-pixel count all-reduced"
                                         + ("; the exchange overlaps the backward pass, the scale goes into the optimizer)" if ddp._defer else ")")
                                         if ddp.exact_batch_loss else "mean of the per-rank masked means"),
                          "call_sequence": "spec (depth_net(cat), slices, photometric_loss)" if args.spec_calls else
                                           "fast path (forward_pair_split + gradient handover)"},
               "timing": "value / ms_per_step: wall clock around EXACTLY `steps` steps between barrier + torch.cuda.synchronize() on both "
                         "sides, max over ranks; nothing but the steps runs in between (no event brackets around the fused op).  The "
                         "region starts on a drained GPU, as the contract's synchronize leaves it: ms_per_step_pipeline_full is the "
                         "mean per step (hip events) of the same number of steps behind 3 lead-in steps with the start event recorded "
                         "WITHOUT draining -- the per-step time of a running loop; ms_per_step_hipevent_median: median over the timed "
                         "region's per-step hip events",
               "first_loss": first_loss, "final_loss": final_loss,
               "deterministic_weight_gradients": bool(dn.deterministic and pn.deterministic),
               "hipgraph": use_graph, "hipgraph_trial": graph_trial, "hipgraph_error": graph_error,
               "hipgraph_policy": ({"policy": args.graph_policy, "group": graphed.capture_group, "carry": bool(graphed.carry),
                                    "graph": graphed.stats} if use_graph else None),
               # what the stream policy saw in this process (a first multi-GPU SCALE record should explain itself: DESIGN.md
               # section 5): the runtime's hardware-queue limit as exported before HIP initialised, queues claimed by parties other
               # than the step's main + weight-gradient stream (RCCL's communicator stream, a loader's copy stream), and how many
               # library-owned auxiliary side streams colvo_run_commands may use as a consequence
               "hw_queues": {"GPU_MAX_HW_QUEUES": _streams.hw_queue_limit(), "external_claims": _streams.external_queues(),
                             "aux_side_streams": _streams.aux_side_streams(), "folded": _streams.folded(),
                             "rccl_env": {k: v for k, v in os.environ.items() if k.startswith(("NCCL_", "RCCL_"))}}}
        out.update(side)
        if not args.spec_calls and "spec_sequence_ms" in side:
            out["spec_sequence_value"] = world * B / (side["spec_sequence_ms"] * 1e-3)
        # `roofline` is the figure SURVEY.md section 8d prescribes: the fused op alone at BASELINE configs[2] (B=32, 640x512), where it
        # is bandwidth- and not latency-dominated.  The same op inside this run's steps: roofline_in_step.
        if not args.no_roofline_cfg2:
            out["roofline"] = roofline_cfg2(dev)
            out["roofline_in_step"] = roof_in_step
        else:
            out["roofline"] = roof_in_step
        fl = conv_flops_per_step(B, H, W)
        peak = 2500.0 if args.dtype == "bf16" else 157.3
        out["mfma_step"] = {"what": "conv flops of the whole step (fwd + dgrad + wgrad, both networks) / step time: a lower "
                                    "bound of the conv kernels' MFMA rate, the step also holds the loss, Adam and packing",
                            "flops_per_step": fl, "achieved": fl / (ms * 1e-3) / 1e12, "peak": peak, "unit": "TFLOP/s",
                            "frac": fl / (ms * 1e-3) / 1e12 / peak}
        if not args.no_cpu_baseline and world == 1:
            out["depth_l1_vs_oracle"] = depth_l1_vs_oracle(dev, cdt, B, H, W)
            out["cpu_baseline"] = cpu_baseline(B, H, W, args.cpu_seconds)
        elif world == 1:
            out["cpu_baseline"] = None
        print(json.dumps(out), flush=True)
    rank_barrier()
    if world > 1 or args.rccl_single:
        torch.cuda.synchronize()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
