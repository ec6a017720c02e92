#!/usr/bin/env python3
"""Instruction counts per basic block of one kernel in a hipcc -S listing (where does the prologue go?):
   tools/isa_segments.py file.s <mangled-name-substring>"""
import re
import sys


def count(lines):
    n = v = s = 0
    for l in lines:
        l = l.strip()
        if not l or l[0] in ";." or l.endswith(":"):
            continue
        n += 1
        v += l.startswith("v_")
        s += l.startswith("s_")
    return n, v, s


def main(path, sub):
    lines = open(path).read().split("\n")
    start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\S*:", l) and sub in l)
    end = next(j for j in range(start, len(lines)) if "s_endpgm" in lines[j])
    print(lines[start][:100], "total (n, valu, salu):", count(lines[start:end]))
    prev, name = start, "entry"
    for j in range(start, end + 1):
        if j == end or re.match(r"^\.LBB\d+_\d+:", lines[j]):
            seg = lines[prev:j]
            c = count(seg)
            mf = sum("v_mfma" in l for l in seg)
            br = [l.strip() for l in seg if re.match(r"\s*s_cbranch|\s*s_branch", l)]
            if c[0] >= 8:
                print(f"{name:14s} n={c[0]:5d} valu={c[1]:5d} salu={c[2]:4d} mfma={mf:3d} {' '.join(b.split()[-1] for b in br)}")
            prev, name = j, lines[j].split(":")[0] if j < end else ""


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
