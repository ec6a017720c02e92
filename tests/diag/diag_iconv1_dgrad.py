"""Diagnostic (GPU box): the iconv1 input-gradient call of the seed-21 golden step against torch's conv backward on the same
tensors."""
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
dn, pn = hnn.DepthNet(compute_dtype=torch.float32), hnn.PoseNet(compute_dtype=torch.float32)
dn.load_state_dict(dn_o.state_dict()); pn.load_state_dict(pn_o.state_dict())
d = {k: v.cuda() for k, v in b.items() if torch.is_tensor(v)}
calls = []
orig = ops.conv_dgrad
def wrapped(desc, src, dy, w_bwd, relu_mask, dx, accumulate):
    orig(desc, src, dy, w_bwd, relu_mask, dx, accumulate)
    calls.append((desc, src, dy, w_bwd, relu_mask, dx, accumulate))
ops.conv_dgrad = wrapped
out = hnn.dcdp_forward(dn, pn, d["tgt"], d["ref"], d["K"])
out[0].backward()
torch.cuda.synchronize()
for (desc, src, dy, w_bwd, mask, dx, acc) in calls:
    if acc:
        continue
    name = f"C0={desc.C0} C1={desc.C1} Cout={desc.Cout} {desc.Hi}x{desc.Wi} s{desc.stride} up{desc.up0}"
    # reference: gradient of conv(x, W) w.r.t. x.  w_bwd is [Cin][9 flipped][Cout]
    Cin = w_bwd.shape[0]
    Wf = w_bwd.float().view(Cin, 3, 3, desc.Cout).flip(1, 2).permute(3, 0, 1, 2).contiguous()   # OIHW
    hs, ws = (desc.Hi // 2, desc.Wi // 2) if desc.up0 else (desc.Hi, desc.Wi)
    x = torch.zeros(desc.B, Cin, desc.Hi, desc.Wi, device="cuda", requires_grad=True)
    y = F.conv2d(x, Wf, stride=desc.stride, padding=1)
    y.backward(dy.float().permute(0, 3, 1, 2))
    ref = x.grad
    if desc.up0:
        ref = F.avg_pool2d(ref, 2) * 4
    ref = ref[:, :desc.C0]
    if mask is not None:
        ref = ref * (mask.float().permute(0, 3, 1, 2) > 0)
    got = dx.float().permute(0, 3, 1, 2)
    err = (got - ref).abs()
    rel = ((got - ref).norm() / ref.norm().clamp_min(1e-30)).item()
    print(f"{name:48s} relL2 {rel:.2e} max err {err.max().item():.2e} (max ref {ref.abs().max().item():.2e})")
    if rel > 1e-4:
        bad = (err > 1e-3 * ref.abs().max()).nonzero()
        print("   bad elements:", len(bad), "first:", bad[:12].tolist())
        ys = bad[:, 2].unique().tolist(); xs = bad[:, 3].unique().tolist(); cs = bad[:, 1].unique().tolist(); bs = bad[:, 0].unique().tolist()
        print("   images", bs, "channels", cs, "rows", ys[:40], "cols", xs[:40])
