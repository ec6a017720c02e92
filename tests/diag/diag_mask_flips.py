"""Diagnostic (GPU box): does the HIP fused loss differ from the oracle on isolated pixels whose validity flips?
   python tests/diag_mask_flips.py seed [B H W]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from coivo_amd import synth, functional as Fh
from oracle import colvo_spec as S

seed = int(sys.argv[1]); B, H, W = (int(v) for v in sys.argv[2:5]) if len(sys.argv) > 4 else (2, 64, 96)
dn, pn = S.make_models(seed)
b = synth.make_batch(B, H, W, seed=seed)
with torch.no_grad():
    d_t, d_r = dn(b["tgt"]), dn(b["ref"])
    pose, a, bb = pn(b["tgt"], b["ref"], d_t, d_r)
leaf = d_t.clone().requires_grad_(True)
lo = S.photometric_loss(b["tgt"], b["ref"], leaf, pose, b["K"], a, bb)
go = torch.autograd.grad(lo, leaf)[0]
x, y, valid = S.project(d_t, pose, b["K"])
dev = torch.device("cuda")
lh_leaf = d_t.clone().to(dev).requires_grad_(True)
lh = Fh.photometric_loss(b["tgt"].to(dev), b["ref"].to(dev), lh_leaf, pose.to(dev), b["K"].to(dev), a.to(dev), bb.to(dev))
gh = torch.autograd.grad(lh, lh_leaf)[0].cpu()
_, vh = Fh.inverse_warp(b["ref"].to(dev), d_t.to(dev), pose.to(dev), b["K"].to(dev))
vh = vh.cpu()
print("loss", lo.item(), lh.item(), "n_valid oracle", valid.sum().item(), "hip", vh.sum().item())
diff = (gh - go).abs()
thr = 1e-3 * go.abs().max()
bad = (diff > thr).nonzero()
print("pixels with |d_depth diff| > 1e-3 max:", len(bad), "of", go.numel(), " relL2", ((gh - go).norm() / go.norm()).item())
vo = valid.view_as(vh) if valid.dim() == vh.dim() else valid.unsqueeze(1)
flip = (vo.float() != vh).nonzero()
print("validity flips:", len(flip))
for i in flip[:10]:
    bi, _, yy, xx = i.tolist()
    print("  flip at", (bi, yy, xx), "x,y =", x.view(B, H, W)[bi, yy, xx].item(), y.view(B, H, W)[bi, yy, xx].item(), "oracle valid", vo[bi, 0, yy, xx].item())
for i in bad[:10]:
    bi, _, yy, xx = i.tolist()
    print("  bad at", (bi, yy, xx), "x,y =", x.view(B, H, W)[bi, yy, xx].item(), y.view(B, H, W)[bi, yy, xx].item(), "go", go[bi, 0, yy, xx].item(), "gh", gh[bi, 0, yy, xx].item())
