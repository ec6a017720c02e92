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


def grad_parity_table(named_hip, named_o32, named_o64, out_path=None):
    """Per-parameter gradient errors of the HIP path and of the fp32 oracle, both measured against the fp64 oracle
    (the yardstick: fp32 summation order alone moves these heavily cancelling sums by up to a few 1e-3 of their largest
    element).  Returns rows (name, n, scale = max|g64|, err_hip = max|hip-g64|, err_o32 = max|o32-g64|,
    l2_hip = |hip-g64|_2 / |g64|_2, l2_o32)."""
    rows = []
    for (n, gh), (_, g32), (_, g64) in zip(named_hip, named_o32, named_o64):
        gh = gh.detach().cpu().double()
        g32 = g32.detach().cpu().double()
        g64 = g64.detach().cpu().double()
        scale = max(g64.abs().max().item(), 1e-30)
        nrm = max(g64.norm().item(), 1e-30)
        rows.append((n, gh.numel(), scale, (gh - g64).abs().max().item(), (g32 - g64).abs().max().item(),
                     (gh - g64).norm().item() / nrm, (g32 - g64).norm().item() / nrm))
    if out_path is not None:
        try:
            with open(out_path, "w") as f:
                f.write("param n scale max_err_hip/scale max_err_o32/scale relL2_hip relL2_o32\n")
                for r in rows:
                    f.write(f"{r[0]} {r[1]} {r[2]:.4e} {r[3] / r[2]:.3e} {r[4] / r[2]:.3e} {r[5]:.3e} {r[6]:.3e}\n")
        except OSError:
            pass
    return rows


# SPEC.md §7: the yardstick for a parameter gradient is the fp64 evaluation of the spec.  These are heavily cancelling
# sums, so the fp32 ORACLE itself sits some way from it -- 2e-6 (relative L2) at B=2 64x96, 5e-4 at B=8 256x320, 2e-3 at
# B=1 256x320 -- and that distance, measured in the same step, is the only scale the bar uses: there is NO absolute floor.
# A gradient tensor passes when
#   * its relative L2 error is at most GRAD_K x the fp32 oracle's on the same tensor, or at most the fp32 oracle's own
#     worst tensor of this step; and
#   * its largest element error (relative to the tensor's largest element) is at most GRAD_K x the fp32 oracle's on the same
#     tensor, or at most GRAD_MAX_SLACK x the fp32 oracle's worst tensor of this step (the maximum over up to 2.4 M elements
#     is an extreme-value statistic: two fp32 summation orders differ by more in it than in the L2 norm).
# Round-3 tables of the final round-2 build (gpurun_out/grad_parity_*.txt): HIP / fp32-oracle worst tensors 2.1e-6 / 1.6e-6
# (B=2 64x96), 7.8e-4 / 4.9e-4 (configs[1]), 2.0e-3 / 1.9e-3 (B=1 256x320); a gradient 3x worse than the oracle's own
# rounding noise fails at every shape.
GRAD_K, GRAD_MAX_SLACK = 3.0, 2.0


def grad_parity_failures(rows):
    worst_l2 = max(r[6] for r in rows)
    worst_max = max(r[4] / r[2] for r in rows)
    bad = []
    for name, n, scale, eh, eo, lh, lo in rows:
        if lh > max(GRAD_K * lo, worst_l2) or eh > max(GRAD_K * eo, GRAD_MAX_SLACK * worst_max * scale):
            bad.append(f"{name}: relL2 hip {lh:.3e} vs o32 {lo:.3e} (worst o32 {worst_l2:.3e}); max err/scale hip "
                       f"{eh / scale:.3e} vs o32 {eo / scale:.3e} (worst o32 {worst_max:.3e})")
    return bad
