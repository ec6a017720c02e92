"""Diagnostic (GPU box): relL2 of every parameter gradient of the coupled step vs the fp64 oracle, HIP and fp32 oracle.
   python tests/diag_golden_grads.py seed [B H W]      (env toggles select kernel variants)"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from coivo_amd import synth, nn as hnn
from oracle import colvo_spec as S

seed = int(sys.argv[1]); B, H, W = (int(v) for v in sys.argv[2:5]) if len(sys.argv) > 4 else (2, 64, 96)
b = synth.make_batch(B, H, W, seed=seed)
res = {}
for tag, dt in (("o32", torch.float32), ("o64", torch.float64)):
    dn, pn = S.make_models(seed, dtype=dt)
    out = S.dcdp_forward(dn, pn, b["tgt"].to(dt), b["ref"].to(dt), b["K"].to(dt))
    out[1].retain_grad(); out[2].retain_grad()
    out[0].backward()
    res[tag] = ([("depth." + n, p.grad) for n, p in dn.named_parameters()] + [("pose." + n, p.grad) for n, p in pn.named_parameters()],
                out[1].grad, out[2].grad)
dn_o, pn_o = S.make_models(seed)
dn, pn = hnn.DepthNet(compute_dtype=torch.float32), hnn.PoseNet(compute_dtype=torch.float32)
dn.load_state_dict(dn_o.state_dict()); pn.load_state_dict(pn_o.state_dict())
d = {k: v.cuda() for k, v in b.items() if torch.is_tensor(v)}
mode = os.environ.get("DIAG_MODE", "fast")
if mode == "fast":
    out = hnn.dcdp_forward(dn, pn, d["tgt"], d["ref"], d["K"])
else:       # spec sequence: general autograd path
    from coivo_amd import functional as Fh
    dd = dn(torch.cat([d["tgt"], d["ref"]]))
    d_t, d_r = dd[:B], dd[B:]
    pose, a, bb = pn(d["tgt"], d["ref"], d_t, d_r)
    out = (Fh.photometric_loss(d["tgt"], d["ref"], d_t, pose, d["K"], a, bb), d_t, d_r)
    d_t.retain_grad(); d_r.retain_grad()
out[0].backward()
torch.cuda.synchronize()
hip = [("depth." + n, p.grad) for n, p in dn.named_parameters()] + [("pose." + n, p.grad) for n, p in pn.named_parameters()]
worst = (0, None)
lines = []
for (n, gh), (_, g32), (_, g64) in zip(hip, res["o32"][0], res["o64"][0]):
    gh = gh.detach().cpu().double(); g64 = g64.double(); g32 = g32.double()
    lh = ((gh - g64).norm() / g64.norm()).item(); lo = ((g32 - g64).norm() / g64.norm()).item()
    lines.append(f"{n:22s} hip {lh:.2e} o32 {lo:.2e}")
    if lh > worst[0]:
        worst = (lh, n)
print(f"seed {seed} mode {mode} env {[k for k in os.environ if k.startswith('COLVO_')]}: worst {worst}")
if mode != "fast":
    for nm, gh, g64 in (("d_t.grad", out[1].grad, res["o64"][1]), ("d_r.grad", out[2].grad, res["o64"][2])):
        print("   ", nm, "relL2 vs o64", ((gh.cpu().double() - g64).norm() / g64.norm()).item())
if os.environ.get("DIAG_VERBOSE"):
    print("\n".join(lines))
