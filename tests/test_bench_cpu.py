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
    assert p["global_batch"] == 8 * n and p["seed"] == 1234 + n - 1 and p["grad_transport"] == "bf16"
    # the keyword set init_process_group accepts
    import inspect
    import torch.distributed as dist
    sig = inspect.signature(dist.init_process_group)
    for k in p["init_kwargs"]:
        assert k in sig.parameters, k


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
              '"scaling"', '"vs_baseline"', '"dtype"', '"data"', '"config"', '"roofline"', '"cpu_baseline"'):
        assert k in src, k
    assert json.dumps({"ok": True})
