"""-m gpu: the hipGraph-captured train step reproduces the eager step (same kernels, same order)."""
import pytest
import torch

from coivo_amd import synth
from tests.gpu_util import dev, to_dev

pytestmark = pytest.mark.gpu


def _run_case(name, *args, env_extra=None):
    """One child process per case (tests/graph_cases.py says why).  A child that dies -- of an assertion or of the runtime's abort()
    -- FAILS the test with its output (round 4 re-ran an aborted child once: removed)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, os.path.join(root, "tests", "graph_cases.py"), name] + [str(a) for a in args]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", **(env_extra or {}))
    r = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=600, cwd=root)
    assert r.returncode == 0 and "GRAPH_CASE_OK " + name in r.stdout, f"rc {r.returncode}\n" + r.stdout[-1500:] + r.stderr[-3000:]


@pytest.mark.parametrize("policy", [0, 1, 2, 3])
@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_graphed_step_matches_eager(dtype, policy):
    """Four replayed steps against four eager steps from the same start, every capture policy; new inputs through the static buffers
    (tests/graph_cases.py case_matches_eager)."""
    _run_case("matches_eager", dtype, policy)


def test_eight_captures_in_one_process_then_a_copy_and_an_eager_step():
    """eight captures one after the other in ONE child process (each replayed against an eager twin, then
    GraphedTrainStep.close()), then the host-to-device copy that round 4's session process aborted in (gpurun_out/r4w/t.log) and an
    eager step; the closed step captures a ninth time.  Run once, no retry.  AMD_LOG_LEVEL=1 so that, should the runtime end the
    child, its own error line is on stderr (round 4's abort was silent)."""
    _run_case("eight_captures_in_one_process", env_extra={"AMD_LOG_LEVEL": "1"})


def test_graphed_step_is_bitwise_the_eager_step_in_deterministic_mode():
    """With deterministic weight gradients nothing in the step depends on an execution order: three replayed steps and three eager
    steps from the same start end in bit-identical parameters, whatever the graph's branch structure (case_bitwise_deterministic)."""
    _run_case("bitwise_deterministic")


def test_graphed_full_objective_step_is_bitwise_the_eager_one():
    """The widened objective inside the captured step, bit for bit the eager trajectory (case_full_objective_bitwise)."""
    _run_case("full_objective_bitwise")


@pytest.mark.parametrize("policy", [0, 2])
def test_graph_is_built_with_explicit_dependencies(policy):
    """The structure of the captured graph read back from the runtime -- one root, a pure chain (policy 0) or two chains with carried
    segments (policy 2), no repacking node (case_explicit_dependencies holds the assertions)."""
    _run_case("explicit_dependencies", policy)


@pytest.mark.parametrize("transport", ["f32", "bf16", "f32-torch", "f32-defer", "bf16-defer"])
def test_graphed_step_with_rccl_single_rank(transport):
    """configs[4]'s structure on one GPU: the graph contains the bucketed all-reduces of ddp.GradBuckets (RCCL, one rank) beside the
    native two-chain backward.  Deterministic weight gradients + fp32 transport: a one-rank all-reduce is the identity, so the
    replayed data-parallel trajectory equals the eager one BIT FOR BIT (losses, both parameter arenas) and both equal the run without
    any process group; with bf16 transport the gradients are rounded to bf16 on the way -- identically in both forms, so that pair is
    bitwise equal too.  One child process per transport (tests/graph_rccl_worker.py): a process group is process-global state, and
    the HIP runtime ends a failed capture with abort() -- seen once in four runs of this test inside the session process (round 4,
    gpurun_out/r4d/t_conv.log: hipStreamEndCapture), which would take every later test down with it."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    # f32 / bf16: RCCL called natively on the group's communicator (ddp._NativeRccl); f32-torch: through ProcessGroup.allreduce (refused
    # first, then with the opt-in); *-defer (round 6): the loss normaliser's exchange started behind the loss kernel, waited for in
    # finish(), the scale applied by the optimizer -- all of it inside the captured step, as `bench.py --config 4` runs data parallel
    transport, path = (transport.split("-") + ["native"])[:2]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0",
               MASTER_PORT={"f32native": "29541", "bf16native": "29542", "f32torch": "29544", "f32defer": "29545", "bf16defer": "29546"}[transport + path])
    r = subprocess.run([sys.executable, os.path.join(root, "tests", "graph_rccl_worker.py"), transport] + ([path] if path != "native" else []),
                       capture_output=True, text=True, env=env, timeout=600, cwd=root)
    assert r.returncode == 0 and f"GRAPH_RCCL_OK {transport}{' DEFER' if path == 'defer' else ''}" in r.stdout, r.stdout[-1500:] + r.stderr[-3000:]


def test_graphed_step_with_rccl_at_the_configs4_per_gpu_shape():
    """the captured data-parallel step at the shape BASELINE configs[4] names per GPU -- 64 pairs of 320x256, bf16,
    RCCL (one rank) inside the graph -- where GraphedTrainStep takes its one-command-segment branch (graph.py: capture_group = 1 from
    32 pairs on) and the conv dispatch picks the large-grid kernel forms: deterministic mode, three replayed steps bit for bit the
    eager data-parallel steps and the steps without a process group (tests/graph_rccl_worker.py)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_PORT="29543")
    r = subprocess.run([sys.executable, os.path.join(root, "tests", "graph_rccl_worker.py"), "f32", "64", "256", "320"],
                       capture_output=True, text=True, env=env, timeout=900, cwd=root)
    assert r.returncode == 0 and "GRAPH_RCCL_OK f32" in r.stdout, f"rc {r.returncode}\n" + r.stdout[-1500:] + r.stderr[-3000:]


def test_capture_and_replay_with_two_hardware_queues_in_a_fresh_process():
    """round 2's multi-stream capture aborted the runtime under GPU_MAX_HW_QUEUES=2.  The capture has used ONE stream
    since round 3; this runs capture + 3 replays + an eager step in a child process whose environment holds GPU_MAX_HW_QUEUES=2
    before it touches the GPU, and checks that the child ends with rc 0 and the eager trajectory's loss."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = (
        "import sys, torch; sys.path.insert(0, %r)\n"
        "from coivo_amd import nn as hnn, synth\n"
        "from coivo_amd.graph import GraphedTrainStep\n"
        "from coivo_amd.optim import FusedAdam\n"
        "dev = torch.device('cuda:0'); B, H, W = 2, 64, 96\n"
        "b = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in synth.make_batch(B, H, W, seed=5).items()}\n"
        "def nets():\n"
        "    torch.manual_seed(0)\n"
        "    dn, pn = hnn.DepthNet(compute_dtype=torch.bfloat16), hnn.PoseNet(compute_dtype=torch.bfloat16)\n"
        "    with torch.no_grad():\n"
        "        for n in (dn, pn):\n"
        "            for name, p in n.named_parameters():\n"
        "                if name.endswith('weight'): p.copy_(torch.randn(p.shape, device=dev) * (2.0 / (p.shape[1] * 9)) ** 0.5)\n"
        "    dn.deterministic = pn.deterministic = True\n"
        "    return dn, pn, FusedAdam([dn, pn], lr=1e-4)\n"
        "dn, pn, opt = nets(); torch.manual_seed(0)\n"
        "step = GraphedTrainStep(dn, pn, opt, B, H, W)\n"
        "frames = torch.cat([b['tgt'], b['ref']])\n"
        "g = [step(frames, b['K']).item() for _ in range(3)]\n"
        "dn2, pn2, opt2 = nets(); e = []\n"
        "for _ in range(3):\n"
        "    opt2.zero_grad(); l = hnn.dcdp_forward(dn2, pn2, b['tgt'], b['ref'], b['K'])[0]; l.backward(); opt2.step(); e.append(l.item())\n"
        "torch.cuda.synchronize()\n"
        "assert g == e, (g, e)\n"
        "print('HWQ2_OK', g[-1])\n") % root
    env = dict(os.environ, GPU_MAX_HW_QUEUES="2", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=600, cwd=root)
    assert r.returncode == 0 and "HWQ2_OK" in r.stdout, r.stdout[-1500:] + r.stderr[-3000:]


def test_eager_steps_after_a_capture_run_at_their_usual_rate():
    """a process that captures the step FIRST (bench.py --graph on, a train loop that captures at start-up) and
    later runs the same networks eagerly -- a validation pass, another batch size -- must not fall off the hardware-queue cliff
    (round 3: 4.6 ms per eager step after a capture against 1.5 in a fresh process; cause and fix: GraphedTrainStep.capture).
    Two child processes at the configs[1] shape: eager steps in a fresh process; capture -> replays -> eager steps."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    tool = os.path.join(root, "tools", "probe_eager_after_capture.py")

    def run(extra):
        r = subprocess.run([sys.executable, tool, "--pairs", "8", "--steps", "30"] + extra, capture_output=True, text=True,
                           timeout=600, cwd=root)
        assert r.returncode == 0, r.stdout[-1000:] + r.stderr[-3000:]
        line = [l for l in r.stdout.splitlines() if l.startswith("PROBE_JSON ")][-1]
        return json.loads(line[len("PROBE_JSON "):])

    fresh = run(["--only-eager"])["eager0"]
    after = run(["--capture-first"])
    assert after["eager_after"] < 1.3 * fresh, (fresh, after)
    assert after["replay"] < 1.3 * fresh, (fresh, after)
