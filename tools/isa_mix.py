#!/usr/bin/env python3
"""Static instruction mix of the kernels in a hipcc -S listing: tools/isa_mix.py file.s [name-substring ...]"""
import collections
import re
import sys


def main(path, filters):
    lines = open(path).read().split("\n")
    starts = [(i, l.split(":")[0]) for i, l in enumerate(lines) if re.match(r"^_Z\S+:", l)]
    for n, (i, name) in enumerate(starts):
        if filters and not any(f in name for f in filters):
            continue
        end = next((j for j in range(i, len(lines)) if "s_endpgm" in lines[j]), len(lines))
        ops = collections.Counter()
        for l in lines[i + 1:end]:
            l = l.strip()
            if not l or l[0] in ";." or l.endswith(":"):
                continue
            ops[l.split()[0]] += 1
        cls = collections.Counter()
        for k, v in ops.items():
            c = ("mfma" if "mfma" in k else "valu" if k.startswith("v_") else "salu" if k.startswith("s_") else
                 "lds" if k.startswith("ds_") else "vmem" if k.startswith(("global_", "buffer_", "flat_", "scratch_")) else "other")
            cls[c] += v
        print(name[-60:], dict(cls))
        print("   ", ops.most_common(40))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2:])
