"""CPU checks of bench.py's launch plumbing: the multi-GPU (RCCL) branch is walked up to -- not including --
init_process_group, so that a typo in it does not surface first on the 8-GPU node (no GPU needed)."""
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _args(argv):
    import bench
    old = sys.argv
    sys.argv = ["bench.py"] + argv
    try:
        return bench.parse()
    finally:
        sys.argv = old


def test_plan_single_gpu_defaults():
    import bench
    a = _args([])
    assert (a.gpus, a.batch_per_gpu, a.height, a.width, a.dtype) == (1, 8, 256, 320, "bf16")
    assert a.config_id == "1" and a.graph == "best" and not a.custom_shape and not a.spec_calls   # one GPU: eager or replay, the faster
    p = bench.plan_distributed(a, {})
    assert p["world"] == 1 and p["backend"] is None and p["device"] == ("cuda", 0) and p["global_batch"] == 8
    assert p["grad_transport"] is None


@pytest.mark.parametrize("n", [2, 4, 8])
def test_plan_rccl_branch(n):
    """The environment `python -m torch.distributed.run --nproc-per-node N` gives rank n-1."""
    import bench
    a = _args(["--gpus", str(n), "--steps", "20", "--warmup", "5", "--grad-transport", "bf16"])
    env = {"RANK": str(n - 1), "LOCAL_RANK": str(n - 1), "WORLD_SIZE": str(n), "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": "29500"}
    p = bench.plan_distributed(a, env)
    assert p["backend"] == "nccl" and p["world"] == n and p["rank"] == n - 1
    assert p["device"] == ("cuda", n - 1) and p["init_kwargs"] == {"rank": n - 1, "world_size": n, "device_id": ("cuda", n - 1)}
    # N > 1 defaults to BASELINE configs[3]: 32 pairs per GPU (batch 256 on 8 GPUs), 320x256
    assert a.config_id == "3" and (a.batch_per_gpu, a.height, a.width) == (32, 256, 320)
    assert p["global_batch"] == 32 * n and p["seed"] == 1234 + n - 1 and p["grad_transport"] == "bf16"
    # the keyword set init_process_group accepts
    import inspect
    import torch.distributed as dist
    sig = inspect.signature(dist.init_process_group)
    for k in p["init_kwargs"]:
        assert k in sig.parameters, k


def test_baseline_configs_map_to_bench_flags():
    """BASELINE.json configs[1..4] -> what bench.py runs (per-GPU pairs, shape, gradient transport, hipGraph)."""
    import bench
    a = _args(["--gpus", "8"])
    assert (a.config_id, a.batch_per_gpu, a.grad_transport, a.graph) == ("3", 32, "f32", "auto")
    assert bench.plan_distributed(a, {"RANK": "0", "LOCAL_RANK": "0", "WORLD_SIZE": "8"})["global_batch"] == 256
    a = _args(["--gpus", "8", "--config", "4"])
    assert (a.batch_per_gpu, a.grad_transport, a.graph) == (64, "bf16", "on")
    assert bench.plan_distributed(a, {"RANK": "3", "LOCAL_RANK": "3", "WORLD_SIZE": "8"})["global_batch"] == 512
    a = _args(["--config", "3"])                       # the scaling denominator: 32 pairs on ONE GPU
    assert (a.gpus, a.batch_per_gpu, a.graph) == (1, 32, "best") and "configs[3]" in bench.workload_name(a, 32, 256, 320)
    assert _args(["--rccl-single"]).graph == "auto"     # the RCCL path is timed eagerly unless the configuration names the graph
    a = _args(["--config", "2"])
    assert (a.batch_per_gpu, a.height, a.width) == (32, 512, 640)
    a = _args(["--gpus", "2", "--batch-per-gpu", "8"])  # explicit flags win and are reported as a custom shape
    assert a.batch_per_gpu == 8 and a.custom_shape and bench.workload_name(a, 8, 256, 320).startswith("custom shape")
    a = _args(["--config", "4", "--graph", "off", "--grad-transport", "f32"])
    assert (a.graph, a.grad_transport) == ("off", "f32")


def test_stream_policy_environment_check():
    """coivo_amd/streams.py: a data-parallel rank WARNS (it does not stop: ADVICE r3) when its environment promises 3..7 hardware
    queues; >= 8 and the folded <= 2 setting are both fine, and COLVO_IGNORE_HW_QUEUES opts out."""
    import warnings
    from coivo_amd import streams
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        assert streams.check_environment(1, {}) is None
        assert streams.check_environment(8, {"GPU_MAX_HW_QUEUES": "8"}) is None
        assert streams.check_environment(8, {"GPU_MAX_HW_QUEUES": "2"}) is None       # folded: measured level with three queues
        assert streams.check_environment(8, {"GPU_MAX_HW_QUEUES": "4", "COLVO_IGNORE_HW_QUEUES": "1"}) is None
    for env in ({}, {"GPU_MAX_HW_QUEUES": "4"}, {"GPU_MAX_HW_QUEUES": "7"}):
        with pytest.warns(RuntimeWarning, match="GPU_MAX_HW_QUEUES"):
            assert "GPU_MAX_HW_QUEUES" in streams.check_environment(8, env)
    assert streams.folded({"GPU_MAX_HW_QUEUES": "2"}) and not streams.folded({}) and not streams.folded({"GPU_MAX_HW_QUEUES": "8"})


def test_bench_does_not_override_an_exported_queue_limit():
    """bench.py sets GPU_MAX_HW_QUEUES=8 for a data-parallel launch only when the user has not exported a value (static check of
    the pre-import block: importing bench in this process must not depend on WORLD_SIZE)."""
    src = open(os.path.join(ROOT, "bench.py")).read()
    head = src[:src.index("import torch\n")]
    assert 'os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")' in head and 'os.environ["GPU_MAX_HW_QUEUES"] =' not in head


def test_stream_policy_claims(monkeypatch):
    """Claims switch the library's auxiliary side stream off, the last release switches it back on (the call into the library
    only stores a number: no GPU needed)."""
    from coivo_amd import streams
    monkeypatch.delenv("GPU_MAX_HW_QUEUES", raising=False)
    streams.reset()
    assert streams.aux_side_streams() == 3 and streams.external_queues() == 0
    c1 = streams.claim_external_queue("rccl")
    c2 = streams.claim_external_queue("loader")
    assert streams.aux_side_streams() == 0 and streams.external_queues() == 2
    c1.release(); c1.release()
    assert streams.aux_side_streams() == 0
    c2.release()
    assert streams.aux_side_streams() == 3 and streams.external_queues() == 0
    assert streams.configure(1) == 0 and streams.configure(0) == 3
    # the networks on ONE shared side stream (nn.share_side_stream, what ddp.GradBuckets arranges): one external queue fits
    assert streams.networks_share_side_stream(True) == 3
    c3 = streams.claim_external_queue("rccl")
    assert streams.aux_side_streams() == 3 and streams.external_queues() == 1
    c4 = streams.claim_external_queue("loader")
    assert streams.aux_side_streams() == 0
    c4.release(); c3.release()
    assert streams.networks_share_side_stream(False) == 3
    streams.reset()
    monkeypatch.setenv("GPU_MAX_HW_QUEUES", "2")
    c = streams.claim_external_queue("x")
    assert streams.aux_side_streams() == 3          # folded onto two queues: the claim changes nothing
    c.release()
    streams.reset()


def test_plan_rejects_mismatched_launch():
    import bench
    with pytest.raises(SystemExit):
        bench.plan_distributed(_args(["--gpus", "8"]), {})                      # forgot torchrun
    with pytest.raises(SystemExit):
        bench.plan_distributed(_args(["--gpus", "4"]), {"RANK": "0", "LOCAL_RANK": "0", "WORLD_SIZE": "8"})


def test_json_contract_keys_are_built():
    """The keys the driver parses are all present in the dictionary bench.py assembles (static check of the source)."""
    src = open(os.path.join(ROOT, "bench.py")).read()
    for k in ('"metric"', '"value"', '"unit"', '"n_gpus"', '"steps"', '"warmup"', '"ms_per_step"', '"higher_is_better"',
              '"scaling"', '"vs_baseline"', '"dtype"', '"data"', '"config"', '"roofline"', '"cpu_baseline"',
              '"single_gpu_same_batch"', '"spec_sequence_ms"', '"frac_real_bytes"', '"roofline_in_step"',
              '"ms_per_step_pipeline_full"', '"forms_interleaved"', '"hw_queues"', '"first_loss"', '"timing"'):
        assert k in src, k
    # the headline's timed steps carry no event brackets around the fused op -- enable_timing() is only ever
    # switched on after the timed region (and inside roofline_cfg2)
    timed = src[src.index("t0 = time.perf_counter()\n    # The headline"):src.index("elapsed = time.perf_counter() - t0")]
    assert "enable_timing" not in timed
    before = src[src.index("def main():"):src.index("t0 = time.perf_counter()\n    # The headline")]
    assert "Fh.enable_timing(" not in before
    assert json.dumps({"ok": True})
