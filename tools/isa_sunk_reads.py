#!/usr/bin/env python3
"""How many MFMAs of a kernel wait for an LDS read issued right in front of them?

hipcc likes to sink every ds_read to just before its first use: "ds_read, s_waitcnt lgkmcnt(0), v_mfma" -- one exposed LDS round trip
per MFMA unless other waves cover it.  Round 5 found that pattern on every MFMA of k_fwd16_head and k_bwd16 (25 / 22 per tile) and
removed it with operand sets requested one step ahead and pinned by sched_barrier (csrc/fwd16.hip, csrc/bwd16.hip, csrc/conv_rt.hip);
in k_wgrad3x3 the pinned order measured SLOWER than hipcc's own, so this is a lead, not a verdict.

   python tools/isa_sunk_reads.py conv.hip 'k_conv3x3INS0_6bf16|k_dgrad'      (source under coivo_amd/csrc, regex on the mangled name)

prints, per matching kernel: MFMA count and how many of them directly follow "ds_read* ; s_waitcnt lgkmcnt(0)" (static counts)."""
import os
import re
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from coivo_amd import build  # noqa: E402


def count(src: str, pat: str = "."):
    """[(mangled kernel name, MFMAs, MFMAs right behind a full LDS drain)] for the kernels of coivo_amd/csrc/<src> matching `pat`."""
    out = f"/tmp/{src[:-4]}.s"
    build.emit_asm(src, out)
    t = open(out).read()
    res = []
    for m in re.finditer(r"^(_Z\S+):\s*; @", t, flags=re.M):
        name = m.group(1)
        if not re.search(pat, name):
            continue
        i = m.end()
        body = [ln.strip() for ln in t[i:t.index("s_endpgm", i)].splitlines()]
        body = [ln for ln in body if ln and not ln.startswith((";", "."))]
        n_mfma = sunk = 0
        for k, ln in enumerate(body):
            if not ln.startswith("v_mfma"):
                continue
            n_mfma += 1
            prev = body[max(0, k - 1)]
            # (a counted wait -- lgkmcnt(N > 0) with younger reads still in flight -- is what a pipeline looks like: only the drain counts)
            if prev.startswith("s_waitcnt") and "lgkmcnt(0)" in prev and any(body[max(0, k - d)].startswith("ds_read") for d in (2, 3)):
                sunk += 1
        res.append((name, n_mfma, sunk))
    return res


def main():
    for name, n_mfma, sunk in count(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else "."):
        print(f"{name[20:120]:100s} mfma {n_mfma:5d}  read-wait-mfma {sunk:4d}  ({100.0 * sunk / max(1, n_mfma):.0f} %)")


if __name__ == "__main__":
    main()
