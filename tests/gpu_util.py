"""Helpers shared by the -m gpu parity tests."""
import numpy as np
import torch


def dev():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return torch.device("cuda:0")


def to_dev(d, keys=None):
    return {k: (v.to(dev()) if torch.is_tensor(v) else v) for k, v in d.items() if keys is None or k in keys}


def assert_close_frac(got, ref, rtol, atol_scale, max_bad_frac, what=""):
    """Element-wise closeness that tolerates a tiny fraction of outliers (pixels whose hard validity
    mask flips under 1-ulp differences of sin/cos between host libm and the GPU)."""
    got = got.detach().cpu().double()
    ref = ref.detach().cpu().double()
    assert got.shape == ref.shape, (what, got.shape, ref.shape)
    atol = atol_scale * max(ref.abs().max().item(), 1e-30)
    bad = (got - ref).abs() > (atol + rtol * ref.abs())
    frac = bad.double().mean().item()
    assert frac <= max_bad_frac, f"{what}: {frac:.3e} of elements differ (allowed {max_bad_frac:.1e}); " \
                                 f"max abs err {(got - ref).abs().max().item():.3e}, atol {atol:.3e}"


def load_npz(path):
    g = np.load(path)
    return {k: torch.from_numpy(np.asarray(g[k])) for k in g.files}
