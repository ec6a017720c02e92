"""-m gpu: 2-rank data-parallel rehearsal of the real HIP networks on the one available GPU (gloo transport), and the
bench.py multi-process code path."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _run(args, timeout=600, extra_env=None):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", **(extra_env or {}))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_port())] + args
    return subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)


def test_two_rank_dp_on_one_gpu_matches_averaged_gradients():
    r = _run([os.path.join(ROOT, "tests", "ddp_gpu_worker.py")])
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    assert "DDP_OK" in r.stdout


def test_two_rank_dp_deterministic_is_bitwise_the_accumulated_batch():
    """Deterministic weight gradients (nn.*.deterministic: per-split slabs + fixed-order second launches instead of float atomics):
    the 2-rank all-reduced gradient, the parameters after Adam and the learned-split-point replay are BITWISE those of one process
    accumulating both ranks' batches (the spec is BatchNorm-free)."""
    r = _run([os.path.join(ROOT, "tests", "ddp_gpu_worker.py")], extra_env={"DDP_DET": "1"})
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    assert "DDP_OK DET" in r.stdout


def test_two_rank_dp_with_the_widened_objective_is_bitwise_too():
    """The same with dcdp_forward(full_loss=True): the objective's own scatter (the geometric term's tap gradients) is fixed-point,
    so data parallel == accumulated batches holds bit for bit for the widened step as well."""
    r = _run([os.path.join(ROOT, "tests", "ddp_gpu_worker.py")], extra_env={"DDP_DET": "1", "DDP_FULL": "1"})
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    assert "DDP_OK DET FULL" in r.stdout


@pytest.mark.parametrize("mode", ["atomics", "deterministic", "oracle"])
def test_two_rank_dp_with_the_loss_normalisation_off_the_critical_path(mode):
    """GradBuckets(defer_loss_normalisation=True, optimizer=...) (round 6): the two-float exchange of the batch's
    valid-pixel count and masked sum is only started behind the loss kernel; the backward pass runs on the unnormalised gradients,
    the buckets sum those, finish() posts world / max(3 n_global, 1) to the optimizer as a device-side factor of its gradient scale.
    Same three checks as the blocking form above: all-reduced gradient and parameters after Adam against one process accumulating
    both ranks' batches with the same arithmetic (bitwise in deterministic mode), and gradient x scale / world, loss and Adam step
    against the ORACLE's big-batch update at the bar of tests/gpu_util.grad_parity_failures."""
    env = {"DDP_DEFER": "1", **({"DDP_DET": "1"} if mode == "deterministic" else {}), **({"DDP_ORACLE": "1"} if mode == "oracle" else {})}
    r = _run([os.path.join(ROOT, "tests", "ddp_gpu_worker.py")], extra_env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    assert ("DDP_OK DET DEFER" if mode == "deterministic" else "DDP_OK DEFER") in r.stdout, r.stdout[-2000:]
    if mode == "oracle":
        assert "DDP_OK ORACLE" in r.stdout, r.stdout[-2000:]


def test_bench_multiprocess_path():
    r = _run([os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--no-cpu-baseline",
              "--no-roofline-cfg2", "--batch-per-gpu", "2", "--height", "64", "--width", "96", "--rehearse-on-one-gpu"])
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == 2 and d["config"]["global_batch"] == 4 and d["value"] > 0 and d["scaling"] == "weak"


def test_two_rank_dp_matches_the_oracle_big_batch_update():
    """SURVEY.md section 8e, "DP == big batch", on the HIP networks against the SPEC (VERDICT r3 missing 2): two ranks share the GPU
    over gloo, fp32 mode; the all-reduced, 1/N-scaled gradient and the parameters after one Adam step are compared with the ORACLE's
    single-process update on the mean of the per-rank losses, at the gradient bar of tests/gpu_util.grad_parity_failures."""
    r = _run([os.path.join(ROOT, "tests", "ddp_gpu_worker.py")], extra_env={"DDP_ORACLE": "1"})
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    assert "DDP_OK ORACLE" in r.stdout, r.stdout[-2000:]


def test_rccl_path_with_one_rank():
    """The REAL multi-GPU code path with world_size 1: init_process_group('nccl'), GradBuckets, all-reduces issued from the
    weight-gradient side stream (bench.py --rccl-single).  The collectives are trivial but the plumbing is not: this is the
    only way to execute it on a one-GPU box.  With deterministic weight gradients (COLVO_DETERMINISTIC=1) nothing in a step depends on
    an execution order and a one-rank fp32 all-reduce is the identity, so the run through the RCCL path must end in EXACTLY the plain
    run's loss (a wrong grad_scale, a dropped bucket or a stale staging copy cannot hide in a tolerance).  bf16 transport rounds
    every gradient element to bf16 before Adam (pinned exactly by the next test; here a sanity bar).  And the step must not fall into the
    serialised mode (several times the plain step time) that too many active hardware queues cause."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_PORT="29533", COLVO_DETERMINISTIC="1")
    common = [sys.executable, os.path.join(root, "bench.py"), "--no-cpu-baseline", "--no-roofline-cfg2", "--no-side-measurements",
              "--steps", "30", "--warmup", "5",
              "--graph", "off"]        # (the one-GPU default `best` runs 23 more steps before the warm-up: different final loss)
    out = {}
    # rccl: RCCL called natively on the group's communicator (ddp._NativeRccl, the default since round 5); rccl_torch: the same step
    # through ProcessGroup.allreduce (COLVO_DDP_TORCH_COLLECTIVES=1), the fallback
    # (--blocking-loss-exchange: the round-5 form, whose arithmetic is the plain run's -- the exactness checks; `rccl_defer`: the default
    #  since round 6, the loss normaliser applied by the optimizer instead of the heads' backward kernels -- same mathematics, other
    #  roundings)
    blk = ["--rccl-single", "--blocking-loss-exchange"]
    for tag, extra, e in (("plain", [], {}), ("rccl", blk, {}), ("rccl_bf16", blk + ["--grad-transport", "bf16"], {}),
                          ("rccl_torch", blk, {"COLVO_DDP_TORCH_COLLECTIVES": "1"}), ("rccl_defer", ["--rccl-single"], {})):
        r = subprocess.run(common + extra, capture_output=True, text=True, env=dict(env, **e), timeout=300, cwd=root)
        assert r.returncode == 0, r.stderr[-2000:]
        out[tag] = json.loads(r.stdout.strip().split("\n")[-1])
        assert out[tag]["deterministic_weight_gradients"]
    assert out["rccl"]["final_loss"] == out["plain"]["final_loss"], (out["rccl"]["final_loss"], out["plain"]["final_loss"])
    assert out["rccl_torch"]["final_loss"] == out["plain"]["final_loss"], (out["rccl_torch"]["final_loss"], out["plain"]["final_loss"])
    assert out["rccl"]["first_loss"] == out["plain"]["first_loss"] == out["rccl_bf16"]["first_loss"] == out["rccl_torch"]["first_loss"]
    assert "natively" in out["rccl"]["config"]["collectives"] and "natively" in out["rccl_bf16"]["config"]["collectives"]
    assert out["rccl_torch"]["config"]["collectives"] == "ProcessGroup.allreduce" and out["plain"]["config"]["collectives"] is None
    # bf16 transport rounds every gradient element to bf16 before Adam.  WHAT it does to the arena is pinned exactly by
    # test_bf16_gradient_transport_is_exactly_a_bf16_rounding below; what that does to a 35-step trajectory is not derivable -- to
    # first order the loss change D = |first_loss - final_loss| moves by O(2^-8 D), but ReLU / validity decisions that flip on the
    # way amplify it (seen: 3.4e-4 and 6.4e-4 on D = 0.017) -- so here only a sanity bar: the same problem, within a tenth of D.
    D = abs(out["plain"]["first_loss"] - out["plain"]["final_loss"])
    assert abs(out["rccl_bf16"]["final_loss"] - out["plain"]["final_loss"]) <= 0.1 * D, \
        (out["rccl_bf16"]["final_loss"], out["plain"]["final_loss"], D)
    assert out["rccl"]["config"]["grad_transport"] == "f32" and out["rccl_bf16"]["config"]["grad_transport"] == "bf16"
    assert "overlaps the backward pass" in out["rccl_defer"]["config"]["batch_loss"] and "overlaps" not in out["rccl"]["config"]["batch_loss"]
    assert out["rccl_defer"]["first_loss"] == out["plain"]["first_loss"]
    assert abs(out["rccl_defer"]["final_loss"] - out["plain"]["final_loss"]) <= 0.02 * D, (out["rccl_defer"]["final_loss"], out["plain"]["final_loss"], D)
    assert out["rccl_defer"]["ms_per_step_hipevent_median"] < 1.6 * out["plain"]["ms_per_step_hipevent_median"]
    assert out["rccl"]["ms_per_step_hipevent_median"] < 1.6 * out["plain"]["ms_per_step_hipevent_median"]
    assert out["rccl_torch"]["ms_per_step_hipevent_median"] < 1.6 * out["plain"]["ms_per_step_hipevent_median"]


def test_bf16_gradient_transport_is_exactly_a_bf16_rounding():
    """One rank over RCCL: the gradient arena after GradBuckets.finish() with bf16 transport holds EXACTLY the fp32 gradients rounded
    to bf16, and with fp32 transport exactly the gradients of the run without a process group (tests/rccl_transport_worker.py):
    every bucket went out, came back and was copied to the right slice -- no tolerance."""
    import subprocess
    import sys
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_PORT="29551")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "rccl_transport_worker.py")], capture_output=True, text=True, env=env,
                       timeout=600, cwd=ROOT)
    assert r.returncode == 0 and "RCCL_TRANSPORT_OK" in r.stdout, r.stdout[-1500:] + r.stderr[-3000:]
