"""-m gpu: BASELINE configs[2] size (B=32, 640x512) end to end -- the whole step runs, stays finite, and the sliced
wgrad path (tensors >= 1 GiB in fp32) equals the unsliced one."""
import pytest
import torch

from coivo_amd import synth
from tests.gpu_util import dev

pytestmark = pytest.mark.gpu


def test_config2_full_step_bf16():
    from coivo_amd import nn as hnn
    from coivo_amd.optim import FusedAdam
    B, H, W = 32, 512, 640
    d = dev()
    dn, pn = hnn.DepthNet(compute_dtype=torch.bfloat16), hnn.PoseNet(compute_dtype=torch.bfloat16)
    g = torch.Generator().manual_seed(0)
    with torch.no_grad():
        for net in (dn, pn):
            for name, p in net.named_parameters():
                if name.endswith("weight"):
                    p.copy_((torch.randn(p.shape, generator=g) * (2.0 / (p.shape[1] * 9)) ** 0.5).to(d))
    b = synth.make_batch(4, H, W, seed=3, device=d)
    rep = lambda t: t.repeat(B // 4, *([1] * (t.dim() - 1))).contiguous()
    tgt, ref, K = rep(b["tgt"]), rep(b["ref"]), rep(b["K"])
    opt = FusedAdam([dn, pn], lr=1e-4)
    losses = []
    for _ in range(2):
        opt.zero_grad()
        loss = hnn.dcdp_forward(dn, pn, tgt, ref, K)[0]
        loss.backward()
        opt.step()
        losses.append(loss.item())
    assert all(0 < l < 1 for l in losses)
    assert torch.isfinite(dn.flat_grad).all() and torch.isfinite(pn.flat_grad).all()
    assert dn.flat_grad.abs().max() > 0


def test_wgrad_batch_slicing_matches_unsliced():
    """fp32 tensors of >= 1 GiB go through the image-sliced path; compare with two explicit half-batch calls."""
    from coivo_amd import ops
    d = dev()
    B, H, W, C = 68, 256, 320, 16          # 68 x 81920 x 16 x 4 B = 1.43 GB > 1 GiB
    x = torch.randn(B, H, W, C, device=d).relu_()
    dy = torch.randn(B, H, W, C, device=d)
    desc = ops.conv_desc(torch.float32, B, H, W, C, C)
    dw = torch.zeros(C, 9, C, device=d); db = torch.zeros(C, device=d)
    ops.conv_wgrad(desc, x, None, dy, dw, db)
    dw2 = torch.zeros_like(dw); db2 = torch.zeros_like(db)
    h = B // 2
    half = ops.conv_desc(torch.float32, h, H, W, C, C)
    ops.conv_wgrad(half, x[:h], None, dy[:h], dw2, db2)
    ops.conv_wgrad(half, x[h:], None, dy[h:], dw2, db2)
    assert torch.allclose(dw, dw2, rtol=1e-4, atol=1e-3 * dw2.abs().max().item())
    assert torch.allclose(db, db2, rtol=1e-4, atol=1e-3 * db2.abs().max().item())
