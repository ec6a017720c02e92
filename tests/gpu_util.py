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
# Tensors of at most GRAD_TINY elements (the heads' biases: 1, 8, 16 values): their L2 norm is a small-sample statistic like
# the maximum -- for the depth head's ONE bias value the two are the same number -- so the maximum's slack applies to it too
# (seen: that scalar at 1.73e-6 against 3 x 5.7e-7 = 1.72e-6 and a worst oracle tensor of 1.61e-6 in one run of several; it is
# summed with float atomics and moves in the last digits from run to run).
GRAD_K, GRAD_MAX_SLACK, GRAD_TINY = 3.0, 2.0, 16


def grad_parity_failures(rows):
    worst_l2 = max(r[6] for r in rows)
    worst_max = max(r[4] / r[2] for r in rows)
    bad = []
    for name, n, scale, eh, eo, lh, lo in rows:
        l2_bar = max(GRAD_K * lo, (GRAD_MAX_SLACK if n <= GRAD_TINY else 1.0) * worst_l2)
        if lh > l2_bar or eh > max(GRAD_K * eo, GRAD_MAX_SLACK * worst_max * scale):
            bad.append(f"{name}: relL2 hip {lh:.3e} vs o32 {lo:.3e} (worst o32 {worst_l2:.3e}); max err/scale hip "
                       f"{eh / scale:.3e} vs o32 {eo / scale:.3e} (worst o32 {worst_max:.3e})")
    return bad


# ---------------------------------------------------------------------------------------------------------------------- #
# Matched ReLU decisions.                                                                                                  #
# A ReLU whose pre-activation lies within rounding distance of zero is decided by the summation order: the HIP kernels and  #
# torch's CPU conv (and the fp32 and fp64 oracle between themselves) may disagree on it, and ONE such element moves every    #
# upstream parameter gradient by ~1e-3 at B=2 64x96 -- a thousand times the rounding noise of everything else (round 3:      #
# tests/golden/net_b2_64x96, pre-activation 4.5e-7 in `up1`; profiles/r3_relu_flip.md).  At 256x320 there are hundreds of    #
# them, which is what the fp32 oracle's own 5e-4 ... 2e-3 distance from its fp64 evaluation consists of.  Gradient parity    #
# is therefore judged against oracles whose ReLU decisions are FORCED to the HIP path's wherever they differ -- allowed only #
# where the oracle's own pre-activation is below RELU_MARGIN in magnitude, anything larger is a genuine forward mismatch     #
# and fails.  With the decisions matched all three evaluations differ by rounding only.                                      #
# ---------------------------------------------------------------------------------------------------------------------- #
RELU_MARGIN = 1e-5


def hip_relu_masks(dn, pn):
    """{'depth.enc1a': bool [2B,C,h,w], ..., 'pose.conv7': ...}: which activations of the LAST recorded forward of the two
    HIP networks are positive (read from the persistent activation buffers of the recorded passes: call right after the step)."""
    masks = {}
    for tag, net in (("depth", dn), ("pose", pn)):
        pools = [p for p in net._insts.values() if p]
        assert len(pools) == 1 and pools[0], "expected exactly one recorded shape (and COLVO_NO_PROGRAM unset)"
        A = pools[0][-1].passes["fwd"][1]
        for k, t in A.items():
            if k == "in":
                continue
            name = k if isinstance(k, str) else f"conv{k}"
            masks[f"{tag}.{name}"] = (t.float().permute(0, 3, 1, 2) > 0).cpu()
    return masks


def oracle_step(seed, batch, dtype=torch.float32, masks=None, weights_seed=None):
    """The oracle's coupled step (spec init `weights_seed`, default = seed) in `dtype`; with `masks` the ReLU decisions are
    forced to them.  -> dict(loss, d_t, d_r, pose, a, b, grads [(name, tensor)], flips, flip_worst)."""
    from oracle import colvo_spec as S
    dn, pn = S.make_models(seed if weights_seed is None else weights_seed, dtype=dtype)
    stats = {"flips": 0, "worst": 0.0}
    if masks is not None:
        for tag, net in (("depth", dn), ("pose", pn)):
            for name, mod in net.named_children():
                m = masks.get(f"{tag}.{name}")
                if m is None:
                    continue

                def hook(mod, inp, out, m=m):
                    flip = (out > 0) != m
                    n = int(flip.sum())
                    if not n:
                        return None
                    stats["flips"] += n
                    stats["worst"] = max(stats["worst"], out[flip].abs().max().item())
                    o = out.detach()
                    # -o has the other sign (exactly); an exact zero becomes +-1e-30.  Added as a CONSTANT: the gradient still
                    # flows through the element, the ReLU behind it now takes the HIP path's decision
                    delta = torch.where(o == 0, torch.where(m, torch.full_like(o, 1e-30), torch.full_like(o, -1e-30)), -2 * o)
                    return out + torch.where(flip, delta, torch.zeros_like(o))
                mod.register_forward_hook(hook)
    t = lambda v: v.to(dtype)
    loss, d_t, d_r, pose, a, b = S.dcdp_forward(dn, pn, t(batch["tgt"]), t(batch["ref"]), t(batch["K"]))
    loss.backward()
    grads = [("depth." + n, p.grad) for n, p in dn.named_parameters()] + [("pose." + n, p.grad) for n, p in pn.named_parameters()]
    return dict(loss=loss.item(), d_t=d_t.detach(), d_r=d_r.detach(), pose=pose.detach(), a=a.detach(), b=b.detach(),
                grads=grads, flips=stats["flips"], flip_worst=stats["worst"])


def matched_grad_rows(seed, batch, dn, pn, out_path=None, weights_seed=None):
    """Gradient parity rows (grad_parity_table) of the HIP networks' current .grad against the fp32 / fp64 oracle evaluated at
    the HIP path's ReLU decisions.  Asserts that decisions were only ever forced at marginal pre-activations."""
    masks = hip_relu_masks(dn, pn)
    o32 = oracle_step(seed, batch, torch.float32, masks, weights_seed)
    o64 = oracle_step(seed, batch, torch.float64, masks, weights_seed)
    for tag, o in (("fp32", o32), ("fp64", o64)):
        assert o["flip_worst"] < RELU_MARGIN, \
            f"a HIP activation has the other sign than the {tag} oracle's pre-activation of magnitude {o['flip_worst']:.3e}: " \
            f"not a rounding-distance ReLU decision but a forward mismatch"
    print(f"matched ReLU decisions: {o32['flips']} forced in the fp32 oracle (largest |pre| {o32['flip_worst']:.2e}), "
          f"{o64['flips']} in the fp64 oracle ({o64['flip_worst']:.2e})")
    hip = [("depth." + n, p.grad) for n, p in dn.named_parameters()] + [("pose." + n, p.grad) for n, p in pn.named_parameters()]
    return grad_parity_table(hip, o32["grads"], o64["grads"], out_path), o32, o64


# ---------------------------------------------------------------------------------------------------------------------- #
# bf16 mode against the oracle.                                                                        #
# What the HIP networks do differently in bf16 mode: weights are rounded to bf16 (operand copies), every conv layer's output #
# is STORED in bf16 (its accumulation is fp32), so is every layer's input gradient, and the packed network inputs; heads,    #
# loss and weight-gradient accumulation stay fp32.  `oracle_step_bf16` evaluates the fp32 oracle on bf16-rounded weights --  #
# the target -- and, with emulate=True, with exactly those roundings inserted: the distance between the two is the size of    #
# the bf16 data path's own noise, measured in the same step on the same inputs, and the bar for the HIP gradients is a       #
# multiple of it (no constant that has to be re-tuned per shape).                                                            #
# ---------------------------------------------------------------------------------------------------------------------- #
class _RoundBf16(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        return x.to(torch.bfloat16).to(x.dtype)

    @staticmethod
    def backward(ctx, g):
        return g.to(torch.bfloat16).to(g.dtype)


def bf16_rounded_state(module):
    return {k: v.to(torch.bfloat16).to(v.dtype) for k, v in module.state_dict().items()}


def oracle_step_bf16(seed, batch, masks=None, emulate=False, weights_seed=None):
    """The fp32 oracle's coupled step on bf16-ROUNDED weights; ReLU decisions forced to `masks` (hip_relu_masks) where given;
    emulate: the bf16 storage points of the HIP networks inserted (see above).  -> dict(loss, d_t, d_r, grads [(name, tensor)])."""
    from oracle import colvo_spec as S
    dn, pn = S.make_models(seed if weights_seed is None else weights_seed)
    dn.load_state_dict(bf16_rounded_state(dn))
    pn.load_state_dict(bf16_rounded_state(pn))
    for tag, net, first, fp32_layers in (("depth", dn, "enc1a", ("head",)), ("pose", pn, "conv1", ("pred",))):
        for name, mod in net.named_children():
            if name in fp32_layers:
                continue
            m = None if masks is None else masks.get(f"{tag}.{name}")

            def hook(mod, inp, out, m=m):
                if emulate:
                    out = _RoundBf16.apply(out)           # (commutes with the ReLU behind it)
                if m is None:
                    return out
                flip = (out > 0) != m
                if not bool(flip.any()):
                    return out
                o = out.detach()
                delta = torch.where(o == 0, torch.where(m, torch.full_like(o, 1e-30), torch.full_like(o, -1e-30)), -2 * o)
                return out + torch.where(flip, delta, torch.zeros_like(o))
            mod.register_forward_hook(hook)
            if emulate and name == first:
                mod.register_forward_pre_hook(lambda mod, inp: (_RoundBf16.apply(inp[0]),))
    loss, d_t, d_r, pose, a, b = S.dcdp_forward(dn, pn, batch["tgt"], batch["ref"], batch["K"])
    loss.backward()
    grads = [("depth." + n, p.grad) for n, p in dn.named_parameters()] + [("pose." + n, p.grad) for n, p in pn.named_parameters()]
    return dict(loss=loss.item(), d_t=d_t.detach(), d_r=d_r.detach(), pose=pose.detach(), grads=grads)
