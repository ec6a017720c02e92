"""Regression probe for the MFMA-result hazard (conv.hip mfma_result_guard): the f32 persistent single-chunk kernel on
4 x 256 x 320 x 16 -> 16, four launches, number of wrong outputs each (must be 0).  python tools/repro_f32_res.py"""
import sys, torch, torch.nn.functional as F
sys.path.insert(0, '.')
from tests import test_conv_gpu as T
from coivo_amd import ops
case = T.CASES[12]; dtype = torch.float32
B, Hi, Wi, C0, C1, up0, up1, Cout, stride = case
g = torch.Generator().manual_seed(hash(case) % (2 ** 31))
rnd = lambda *s: torch.randn(*s, generator=g)
x0 = F.relu(rnd(B, C0, Hi, Wi)); w = (rnd(Cout, 9, C0) * (2.0 / (9 * C0)) ** 0.5); bias = torch.randn(Cout, generator=g) * 0.1
ref = T._ref_conv(x0, None, False, False, w, bias, 1, True)
d = torch.device('cuda:0')
desc = ops.conv_desc(dtype, B, Hi, Wi, C0, Cout, stride=1, relu=True)
x0d = T._nhwc(x0).to(d); wd = w.to(d); bd = bias.to(d)
nwarm = int(sys.argv[1]) if len(sys.argv) > 1 else 0
for _ in range(nwarm):    # other kernels first
    T.test_conv_fwd_dgrad_wgrad(T.CASES[11], torch.bfloat16)
for trial in range(4):
    yd = torch.full((B, desc.Ho, desc.Wo, Cout), 777.0, device=d)
    ops.conv_fwd(desc, x0d, None, wd, bd, yd)
    torch.cuda.synchronize()
    y = T._nchw(yd).cpu()
    bad = (y - ref).abs() > 1e-3
    print('trial', trial, 'bad', int(bad.sum()))
