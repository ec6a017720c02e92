#!/usr/bin/env python3
"""tools/bench_conv.py log -> the tracked per-layer table:
   python tools/conv_layers_csv.py gpurun_out/r4final/conv16.log "16 frames of 256x320 (BASELINE configs[1]: 8 pairs)" > profiles/r4_conv_layers.csv"""
import re
import sys

log, what = sys.argv[1], sys.argv[2]
# optional third argument: the vendor-library yardstick of the same layers (tools/bench_conv_miopen.py CSV) -> miopen_* columns
mi, mi_note = {}, ""
if len(sys.argv) > 3:
    for line in open(sys.argv[3]):
        if line.startswith("#"):
            if "torch" in line:
                mi_note = line[1:].strip()
            continue
        f = line.strip().split(",")
        if len(f) == 5 and f[0] != "layer":
            mi[f[0]] = tuple(float(v) for v in f[2:5])
print(f"# tools/bench_conv.py on MI355X ({log}): DepthNet conv stack, {what}, bf16; each kernel timed alone (hip events, 30 launches after "
      "warm-up).")
print("# GFLOP = 2*Cout*Cin*9*Ho*Wo*B; MFMA rate = GFLOP / us * 1000 / 2500 TFLOP/s (dense bf16 peak).  iconv1 here is the plain "
      "three-kernel form; the training step runs it fused (k_fwd16_head / k_bwd16, csrc/fwd16.hip, csrc/bwd16.hip)")
if mi:
    print(f"# miopen_*: the same layer through torch / MIOpen ({mi_note}; tools/bench_conv_miopen.py): the bare convolution on a "
          "MATERIALISED input (up-sampling and skip concat are free for it; its input gradient of an up-sampled layer stops at the "
          "up-sampled tensor) -- a lower bound of the library path.  '<' marks the passes where the hand kernel loses.")
print("layer,shape,gflop,fwd_us,fwd_mfma_frac,dgrad_us,dgrad_mfma_frac,wgrad_us,wgrad_mfma_frac" +
      (",miopen_fwd_us,miopen_dgrad_us,miopen_wgrad_us,hand_loses" if mi else ""))
tot = [0.0, 0.0, 0.0, 0.0]
for line in open(log):
    m = re.match(r"(\w+)\s+(.+?)\s+([\d.]+) \|\s+([\d.]+)\s+[\d.]+ \|\s+([\d.]+)\s+[\d.]+ \|\s+([\d.]+)\s+[\d.]+\s*$", line)
    if not m:
        continue
    name, shape, gf, f, d, w = m.group(1), " ".join(m.group(2).split()), float(m.group(3)), float(m.group(4)), float(m.group(5)), float(m.group(6))
    fr = lambda us: gf / us * 1000.0 / 2500.0
    extra = ""
    if name in mi:
        mf, md, mw = mi[name]
        lose = "".join(t for t, ours, theirs in (("fwd<", f, mf), ("dgrad<", d, md), ("wgrad<", w, mw)) if ours > theirs)
        extra = f",{mf:.1f},{md:.1f},{mw:.1f},{lose or '-'}"
        mtot = [a + b for a, b in zip(mtot, (mf, md, mw))] if "mtot" in dir() else [mf, md, mw]
    print(f"{name},{shape},{gf:.2f},{f:.1f},{fr(f):.3f},{d:.1f},{fr(d):.3f},{w:.1f},{fr(w):.3f}{extra}")
    tot = [tot[0] + gf, tot[1] + f, tot[2] + d, tot[3] + w]
for line in open(log):
    if line.startswith("totals:") or line.startswith("MFMA rate per pass"):
        print("# " + line.strip())
if mi and "mtot" in dir():
    print(f"# MIOpen totals: fwd {mtot[0]:.0f} us  dgrad {mtot[1]:.0f} us  wgrad {mtot[2]:.0f} us")
