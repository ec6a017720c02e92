#!/usr/bin/env python3
"""Summarise gpurun_out/pmc_conv_<tag>/ (tools/pmc_conv.sh) into profiles/<tag>_conv_pmc.json: per conv kernel dispatch
shape (kernel name x grid), the mean counter values, duration, and the ratios that say what bounds the kernel.
   tools/pmc_conv_summary.py <tag> [frames] [out.json]"""
import collections
import csv
import glob
import json
import os
import sys

TAG = sys.argv[1] if len(sys.argv) > 1 else "r2"
FRAMES = sys.argv[2] if len(sys.argv) > 2 else "16"
OUTNAME = sys.argv[3] if len(sys.argv) > 3 else f"profiles/{TAG}_conv_pmc.json"
SRC = f"gpurun_out/pmc_conv_{TAG}"


def short(n):
    return n.replace("colvo::(anonymous namespace)::", "").replace("void ", "").split("(")[0]


acc = collections.defaultdict(lambda: collections.defaultdict(list))
for grp in ("sq1", "sq2", "tcp", "tcc"):
    fs = glob.glob(f"{SRC}/{grp}/*/*_counter_collection.csv")
    if not fs:
        continue
    for r in csv.DictReader(open(fs[0])):
        k = (short(r["Kernel_Name"]), int(r["Grid_Size"]), int(r.get("LDS_Block_Size", 0) or 0))
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    ft = glob.glob(f"{SRC}/{grp}/*/*_kernel_trace.csv")
    if ft and grp == "sq1":
        for r in csv.DictReader(open(ft[0])):
            k = (short(r["Kernel_Name"]), int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"]), int(r.get("LDS_Block_Size", 0) or 0))
            acc[k]["duration_ns"].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))

out = []
for (name, grid, lds), c in sorted(acc.items(), key=lambda kv: -sum(kv[1].get("duration_ns", [0]))):
    if not name.startswith(("k_conv", "k_wgrad", "k_dgrad")):       # every conv-stack kernel form
        continue
    m = {k: sum(v) / len(v) for k, v in c.items()}
    if not m.get("duration_ns"):
        continue          # seen in a counter pass only (dispatch shapes of the sq1 pass define the table)
    d = {"kernel": name, "grid_threads": grid, "lds_bytes": lds, "launches": len(c.get("duration_ns", [])),
         "duration_us": m.get("duration_ns", 0) / 1e3}
    wc = m.get("SQ_WAVE_CYCLES")
    if wc:
        d["wave_cycles_share"] = {"wait_any(waitcnt/barrier)": m.get("SQ_WAIT_ANY", 0) / wc,
                                  "wait_inst_any(issue stall)": m.get("SQ_WAIT_INST_ANY", 0) / wc,
                                  "active_inst_any": m.get("SQ_ACTIVE_INST_ANY", 0) / wc,
                                  "active_inst_lds": m.get("SQ_ACTIVE_INST_LDS", 0) / wc,
                                  "active_inst_vmem": m.get("SQ_ACTIVE_INST_VMEM", 0) / wc}
    if m.get("SQ_BUSY_CYCLES") and m.get("SQ_VALU_MFMA_BUSY_CYCLES") is not None:
        # SQ_VALU_MFMA_BUSY_CYCLES counts cycles summed over SIMDs; SQ_BUSY_CYCLES is per SE-ish: report the raw ratio and
        # the absolute MFMA-busy time per SIMD
        d["mfma_busy_cycles"] = m["SQ_VALU_MFMA_BUSY_CYCLES"]
        d["mfma_busy_us_per_simd_at_2.4GHz"] = m["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024.0 / 2400.0
        d["mfma_busy_frac_of_duration"] = d["mfma_busy_us_per_simd_at_2.4GHz"] / max(d["duration_us"], 1e-9)
    if m.get("SQ_LDS_IDX_ACTIVE") is not None:
        d["lds_idx_active_cycles"] = m["SQ_LDS_IDX_ACTIVE"]
        d["lds_bank_conflict_cycles"] = m.get("SQ_LDS_BANK_CONFLICT", 0)
        d["lds_conflict_frac"] = m.get("SQ_LDS_BANK_CONFLICT", 0) / max(m["SQ_LDS_IDX_ACTIVE"], 1)
        d["lds_active_us_per_cu_at_2.4GHz"] = m["SQ_LDS_IDX_ACTIVE"] / 256.0 / 2400.0
        d["lds_active_frac_of_duration"] = d["lds_active_us_per_cu_at_2.4GHz"] / max(d["duration_us"], 1e-9)
    if m.get("TCP_TCC_READ_REQ_sum"):
        d["l1_to_l2_read_req"] = m["TCP_TCC_READ_REQ_sum"]
        d["l1_to_l2_read_latency_cycles_per_req"] = m.get("TCP_TCC_READ_REQ_LATENCY_sum", 0) / m["TCP_TCC_READ_REQ_sum"]
        d["ta_busy_cycles_sum"] = m.get("TA_TA_BUSY_sum")
        d["tcp_pending_stall_cycles_sum"] = m.get("TCP_PENDING_STALL_CYCLES_sum")
    if m.get("TCC_REQ_sum"):
        d["l2_hit_rate"] = m.get("TCC_HIT_sum", 0) / max(m.get("TCC_HIT_sum", 0) + m.get("TCC_MISS_sum", 0), 1)
        d["l2_ea_rdreq"] = m.get("TCC_EA0_RDREQ_sum")
    d["raw"] = {k: v for k, v in m.items() if k != "duration_ns"}
    out.append(d)

os.makedirs("profiles", exist_ok=True)
json.dump({"source": f"rocprofv3 --kernel-trace --pmc <group> -- python3 tools/bench_conv.py {FRAMES} bf16 (tools/pmc_conv.sh {TAG} {FRAMES}); "
                     "one pass per counter group; durations from the sq1 pass (profiled: slower than un-profiled runs)",
           "kernels": out}, open(OUTNAME, "w"), indent=1)
for d in out[:40]:
    w = d.get("wave_cycles_share", {})
    print(f"{d['kernel'][:52]:52s} grid {d['grid_threads']:7d} {d['duration_us']:7.1f} us  wait {w.get('wait_any(waitcnt/barrier)', 0):.2f} "
          f"stall {w.get('wait_inst_any(issue stall)', 0):.2f} act {w.get('active_inst_any', 0):.2f} | mfma {d.get('mfma_busy_frac_of_duration', 0):.2f} "
          f"lds {d.get('lds_active_frac_of_duration', 0):.2f} confl {d.get('lds_conflict_frac', 0):.2f} | L2 lat {d.get('l1_to_l2_read_latency_cycles_per_req', 0):.0f} hit {d.get('l2_hit_rate', 0):.2f}")
