#!/usr/bin/env python3
"""tools/bench_conv.py log -> the tracked per-layer table:
   python tools/conv_layers_csv.py gpurun_out/r4final/conv16.log "16 frames of 256x320 (BASELINE configs[1]: 8 pairs)" > profiles/r4_conv_layers.csv"""
import re
import sys

log, what = sys.argv[1], sys.argv[2]
print(f"# tools/bench_conv.py on MI355X ({log}): DepthNet conv stack, {what}, bf16; each kernel timed alone (hip events, 30 launches after "
      "warm-up).")
print("# GFLOP = 2*Cout*Cin*9*Ho*Wo*B; MFMA rate = GFLOP / us * 1000 / 2500 TFLOP/s (dense bf16 peak).  iconv1 here is the plain "
      "three-kernel form; the training step runs it fused (k_fwd16_head / k_bwd16, csrc/fwd16.hip, csrc/bwd16.hip)")
print("layer,shape,gflop,fwd_us,fwd_mfma_frac,dgrad_us,dgrad_mfma_frac,wgrad_us,wgrad_mfma_frac")
tot = [0.0, 0.0, 0.0, 0.0]
for line in open(log):
    m = re.match(r"(\w+)\s+(.+?)\s+([\d.]+) \|\s+([\d.]+)\s+[\d.]+ \|\s+([\d.]+)\s+[\d.]+ \|\s+([\d.]+)\s+[\d.]+\s*$", line)
    if not m:
        continue
    name, shape, gf, f, d, w = m.group(1), " ".join(m.group(2).split()), float(m.group(3)), float(m.group(4)), float(m.group(5)), float(m.group(6))
    fr = lambda us: gf / us * 1000.0 / 2500.0
    print(f"{name},{shape},{gf:.2f},{f:.1f},{fr(f):.3f},{d:.1f},{fr(d):.3f},{w:.1f},{fr(w):.3f}")
    tot = [tot[0] + gf, tot[1] + f, tot[2] + d, tot[3] + w]
for line in open(log):
    if line.startswith("totals:") or line.startswith("MFMA rate per pass"):
        print("# " + line.strip())
