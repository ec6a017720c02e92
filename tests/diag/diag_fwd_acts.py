"""Diagnostic (GPU box): DepthNet forward activations, HIP fp32 vs the fp32 oracle, layer by layer (seed-21 golden step)."""
import os, sys
os.environ["COLVO_NO_PROGRAM"] = "1"
import torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from coivo_amd import synth, nn as hnn, ops
from oracle import colvo_spec as S

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 21
B, H, W = 2, 64, 96
b = synth.make_batch(B, H, W, seed=seed)
dn_o, pn_o = S.make_models(seed)
acts_o = {}
for name, m in dn_o.named_children():
    m.register_forward_hook(lambda mod, inp, out, name=name: acts_o.__setitem__(name, out.detach()))
with torch.no_grad():
    x = torch.cat([b["tgt"], b["ref"]])
    do = dn_o(x)
dn = hnn.DepthNet(compute_dtype=torch.float32)
dn.load_state_dict(dn_o.state_dict())
outs = []
orig = ops.conv_fwd
def wrapped(desc, x0, x1, w, bias, y):
    orig(desc, x0, x1, w, bias, y)
    outs.append(y)
ops.conv_fwd = wrapped
with torch.no_grad():
    dh = dn(x.cuda())
torch.cuda.synchronize()
names = [f"enc{i}{s}" for i in range(1, 6) for s in "ab"] + [n for i in range(5, 0, -1) for n in (f"up{i}", f"iconv{i}")]
print("depth max abs diff", (dh.cpu() - do).abs().max().item())
for n, y in zip(names, outs):
    yo = F.relu(acts_o[n])
    yh = y.float().permute(0, 3, 1, 2).cpu()
    diff = (yh - yo).abs()
    flips = ((yh > 0) != (yo > 0))
    tiny = (acts_o[n].abs() < 1e-6 * acts_o[n].abs().max()).float().mean().item()
    print(f"{n:8s} max|y| {yo.max().item():.3e} max diff {diff.max().item():.2e} relL2 {((yh - yo).norm() / yo.norm()).item():.2e} "
          f"mask flips {int(flips.sum())} of {flips.numel()}  |pre|<1e-6max: {tiny:.2e}")
    if int(flips.sum()) > 0 and n in ("up1", "up2"):
        idx = flips.nonzero()[:8]
        for i in idx:
            i = tuple(i.tolist())
            print("     flip", i, "oracle pre", acts_o[n][i].item(), "hip y", yh[i].item())
