"""CPU-side checks of the boundary: the C-ABI library builds, loads next to torch's HIP runtime and exports every
symbol include/colvo.h declares; argument validation fails loudly without touching a GPU."""
import ctypes as C
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    from coivo_amd import _lib, build
    build.build()
    return _lib.load()


def _declared_symbols():
    src = open(os.path.join(ROOT, "include", "colvo.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(colvo_[a-z0-9_]+)\s*\(", src)))


def test_header_symbols_are_exported_and_bound(lib):
    from coivo_amd import _lib
    names = _declared_symbols()
    assert len(names) >= 19
    for n in names:
        assert hasattr(lib, n), f"libcolvo.so does not export {n}"
        assert n in _lib.SIGNATURES, f"{n} is declared in colvo.h but has no ctypes signature"
    assert sorted(_lib.SIGNATURES) == names


def test_abi_version_and_error_string(lib):
    from coivo_amd import _lib
    assert lib.colvo_abi_version() == _lib.ABI_VERSION
    assert isinstance(lib.colvo_last_error(), bytes)


def test_workspace_size_formula(lib):
    # max(backward: 14 partial sums per 60-column strip segment, forward: 2 per 62-column strip segment; >= 4 rows each)
    assert lib.colvo_warp_loss_workspace_floats(8, 256, 320) == max(8 * 6 * 64 * 16, 8 * 6 * 64 * 2)   # 16 = 14 gradient sums + loss + count (fused pass)
    assert lib.colvo_warp_loss_workspace_floats(2, 33, 47) == max(2 * 1 * 9 * 16, 2 * 1 * 9 * 2)
    assert lib.colvo_warp_loss_workspace_floats(0, 10, 10) == 0


def test_argument_validation_fails_loudly_without_gpu(lib):
    # null pointers / bad shapes are rejected before any launch
    rc = lib.colvo_warp_loss_fwd(0, 0, 0, 0, 0, 0, 0, 1, 8, 8, 0.85, 0, 0, 0)
    assert rc != 0 and b"null pointer" in lib.colvo_last_error()
    buf = (C.c_float * 16)()
    p = C.addressof(buf)
    rc = lib.colvo_warp_loss_fwd(p, p, p, p, p, p, p, 1, 1, 8, 0.85, p, p, 0)   # H = 1 < 2
    assert rc != 0 and b"bad shape" in lib.colvo_last_error()
    from coivo_amd._lib import ConvDesc
    d = ConvDesc()
    d.dtype, d.B, d.Hi, d.Wi, d.Ho, d.Wo, d.Cout, d.ksize, d.stride, d.C0 = 0, 1, 8, 8, 8, 8, 16, 5, 1, 16
    rc = lib.colvo_conv_fwd(C.byref(d), p, 0, p, p, p, 0)
    assert rc != 0 and b"3x3" in lib.colvo_last_error()
    d.ksize, d.C0 = 3, 12
    rc = lib.colvo_conv_fwd(C.byref(d), p, 0, p, p, p, 0)
    assert rc != 0 and b"multiples of 8" in lib.colvo_last_error()


def test_round3_entry_points_validate_their_arguments(lib):
    """The widened objective, the multi-arena optimizer calls: rejected before any launch, with a message."""
    buf = (C.c_float * 64)()
    p = C.addressof(buf)
    # workspace size: 0 for shapes the call would refuse (H, W not divisible by 2^(scales-1); more than 4 scales)
    assert lib.colvo_full_objective_workspace_floats(2, 48, 64, 3) > 0
    assert lib.colvo_full_objective_workspace_floats(2, 48, 62, 3) == 0
    assert lib.colvo_full_objective_workspace_floats(2, 48, 64, 5) == 0
    assert lib.colvo_full_objective_workspace_floats(2, 48, 64, 1) < lib.colvo_full_objective_workspace_floats(2, 48, 64, 3)
    rc = lib.colvo_full_objective_fwd(p, p, p, 0, p, p, p, p, 2, 48, 64, 3, 0.85, 0.5, 0.1, p, p, 0)      # geo term without depth_r
    assert rc != 0 and b"reference depth" in lib.colvo_last_error()
    rc = lib.colvo_full_objective_fwd(p, p, p, p, p, p, p, p, 2, 48, 62, 3, 0.85, 0.5, 0.1, p, p, 0)
    assert rc != 0 and b"bad shape" in lib.colvo_last_error()
    rc = lib.colvo_full_objective_fwd(p, p, p, p, p, p, p, p, 2, 48, 64, 3, 0.85, 0.5, 0.1, p + 4, p, 0)  # misaligned workspace
    assert rc != 0 and b"16-byte aligned" in lib.colvo_last_error()
    rc = lib.colvo_full_objective_bwd(p, p, p, 2, 48, 64, 3, 0.5, 0.1, p, 0, p, p, p, 0)                   # geo term without d_depth_r
    assert rc != 0 and b"d_depth_r" in lib.colvo_last_error()
    rc = lib.colvo_adam_step_multi(0, 2, 1e-3, 0.9, 0.999, 1e-8, 1.0, 1, 0)
    assert rc != 0 and b"colvo_adam_step_multi" in lib.colvo_last_error()
    from coivo_amd._lib import AdamArena
    arr = (AdamArena * 1)()
    arr[0].param, arr[0].grad, arr[0].exp_avg, arr[0].exp_avg_sq, arr[0].n = p + 4, p, p, p, 8              # misaligned arena
    rc = lib.colvo_adam_step_multi(arr, 1, 1e-3, 0.9, 0.999, 1e-8, 1.0, 1, 0)
    assert rc != 0 and b"16-byte aligned" in lib.colvo_last_error()
    rc = lib.colvo_adam_step_multi(arr, 5, 1e-3, 0.9, 0.999, 1e-8, 1.0, 1, 0)                               # more than COLVO_MAX_ARENAS
    assert rc != 0
    ptrs, sizes = (C.c_void_p * 1)(p), (C.c_size_t * 1)(20)                                                  # not a multiple of 16 bytes
    rc = lib.colvo_zero_multi(ptrs, sizes, 1, 0)
    assert rc != 0 and b"multiple of 16" in lib.colvo_last_error()
    rc = lib.colvo_adam_pack_step(0, 0, 1, 1, 1e-3, 0.9, 0.999, 1e-8, 1.0, 0, 1, 0)                         # no table
    assert rc != 0 and b"colvo_adam_pack_step" in lib.colvo_last_error()


def test_python_ops_refuse_cpu_tensors():
    from coivo_amd import functional as Fh
    t = torch.zeros(1, 3, 8, 8)
    with pytest.raises(RuntimeError, match="GPU only"):
        Fh.photometric_loss(t, t, torch.ones(1, 1, 8, 8), torch.zeros(1, 6), torch.eye(3)[None], torch.ones(1, 1),
                            torch.zeros(1, 1))
    with pytest.raises(RuntimeError, match="GPU only"):
        Fh.inverse_warp(t, torch.ones(1, 1, 8, 8), torch.zeros(1, 6), torch.eye(3)[None])


def test_product_package_does_not_import_the_oracle():
    """The oracle is test infrastructure: nothing under coivo_amd/ may reference it."""
    pkg = os.path.join(ROOT, "coivo_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                txt = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", txt, flags=re.M), f
                assert "colvo_spec" not in txt or f in ("nn.py", "functional.py", "optim.py", "inference.py", "data.py", "frames.hip", "misc.hip",
                                                       "conv.hip", "conv_rt.hip", "wgrad_rt.hip", "warp_loss.hip", "reconstruct.hip"), f   # docstring citations only
    for f in ("nn.py", "functional.py", "optim.py", "ops.py", "ddp.py", "synth.py", "_lib.py", "build.py"):
        txt = open(os.path.join(pkg, f)).read()
        assert not re.search(r"^\s*(from|import)\s+oracle", txt, flags=re.M)


def test_tuning_table_hooks_and_no_stray_getenv(lib):
    """One table of dispatch thresholds (csrc/tuning.h): entries can be read / set by name, unknown names fail loudly, and no
    kernel source reads an environment variable of its own (the developer-build ablation switches excepted)."""
    from coivo_amd import _lib
    assert _lib.tune_get("bn64_min_wgs") == 4096 and _lib.tune_get("wgrad_atomic_mb") == 3
    _lib.tune_set("quad_min_wgs", 7)
    assert _lib.tune_get("quad_min_wgs") == 7
    _lib.tune_set("quad_min_wgs", 2048)
    assert lib.colvo_tune_set(b"no_such_entry", 1.0) != 0 and b"no tuning entry" in lib.colvo_last_error()
    csrc = os.path.join(ROOT, "coivo_amd", "csrc")
    for f in os.listdir(csrc):
        if f == "tuning.h":
            continue
        for line in open(os.path.join(csrc, f)):
            if "getenv(" in line:
                # (developer builds only: -DCOLVO_ABLATE of conv.hip, -DCOLVO_WTRACE of wgrad.hip; never the production library)
                    assert "COLVO_ABL" in line or "COLVO_TRACE" in line or "COLVO_WTRACE" in line, f"{f}: {line.strip()}"
    # the Python side honours its developer switches only under COLVO_DEV=1
    os.environ["COLVO_TEST_SWITCH"] = "x"
    dev = os.environ.pop("COLVO_DEV", None)
    try:
        assert _lib.dev_env("COLVO_TEST_SWITCH") is None
        os.environ["COLVO_DEV"] = "1"
        assert _lib.dev_env("COLVO_TEST_SWITCH") == "x"
    finally:
        os.environ.pop("COLVO_TEST_SWITCH")
        os.environ.pop("COLVO_DEV", None)
        if dev is not None:
            os.environ["COLVO_DEV"] = dev
