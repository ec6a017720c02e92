#!/usr/bin/env python3
"""Static check of the MFMA accumulator chains in a hipcc -S listing (gfx950).

Background (profiles/r2_mfma_hazard.md, tools/ubench/mfma_hazard.hip): an MFMA that accumulates IN PLACE (vDst == SrcC)
is covered by hipcc's wait-state tables (and, for v_mfma_f32_16x16x4_f32, by a hardware interlock), but hipcc's register
allocator sometimes ROTATES the accumulators of a chain (vDst != SrcC).  Measured on MI355X: the result of such a
rotated MFMA needs >= 10 (f32 16x16x4) / >= 8 (bf16 16x16x32) wait states before a VALU read, the bf16 figure being one
MORE than the 7 hipcc inserts.  This tool finds every rotated MFMA and counts the wait states up to the first
non-MFMA instruction that reads its destination.

    tools/isa_check_mfma.py file.s [--min-f32 N] [--min-bf16 N] [--kernel substr]   -> exit code 1 on a violation
"""
import re
import sys

REG = re.compile(r"\b([av])\[(\d+):(\d+)\]|\b([av])(\d+)\b")


def regs(tok):
    """Set of (file, index) an operand token names."""
    out = set()
    for m in REG.finditer(tok):
        if m.group(1):
            out |= {(m.group(1), i) for i in range(int(m.group(2)), int(m.group(3)) + 1)}
        else:
            out.add((m.group(4), int(m.group(5))))
    return out


def parse(line):
    line = line.split(";")[0].strip()
    if not line or line.endswith(":") or line.startswith("."):
        return None
    parts = line.split(None, 1)
    ops = [o.strip() for o in parts[1].split(",")] if len(parts) > 1 else []
    return parts[0], ops


def wait_states(op, ops):
    if op == "s_nop":
        return int(ops[0], 0) + 1
    return 1


def scan(body, labels, pos, dst, ws, need, depth=0):
    """Follow the control flow from body[pos] until `need` wait states have passed: returns None when no non-MFMA
    instruction reads `dst` before that, else (what, wait states)."""
    while pos < len(body):
        if ws >= need or depth > 12:
            return None
        q = parse(body[pos])
        pos += 1
        if not q:
            continue
        op2, ops2 = q
        if op2 == "s_endpgm":
            return None
        if op2 == "s_branch":
            return scan(body, labels, labels[ops2[0]], dst, ws + 1, need, depth + 1) if ops2[0] in labels else None
        if op2.startswith("s_cbranch"):
            if ops2[0] in labels:
                r = scan(body, labels, labels[ops2[0]], dst, ws + 1, need, depth + 1)
                if r:
                    return r
            ws += 1
            continue
        is_store = op2.startswith(("ds_write", "ds_store", "global_store", "buffer_store", "global_atomic", "buffer_atomic"))
        srcs = set()
        for o in (ops2 if is_store else ops2[1:]):
            srcs |= regs(o)
        if op2.startswith("v_mfma"):
            d2, c2 = regs(ops2[0]), regs(ops2[3])
            if d2 == dst and c2 == dst:
                return None                      # an in-place successor: the interlocked / table-covered form takes over
            if (d2 & dst) and not (c2 & dst):
                return None                      # overwritten
            ws += 1                              # counted as ONE state (it really holds the matrix pipe for >= 4 passes)
            continue
        if srcs & dst:
            return ("read by " + op2, ws)
        if not is_store and ops2 and (regs(ops2[0]) & dst):
            dst = dst - regs(ops2[0])            # (partly) overwritten by something else
            if not dst:
                return None
        ws += wait_states(op2, ops2)
    return None


def check(path, min_f32, min_bf16, only=None):
    lines = open(path).read().split("\n")
    starts = [(i, l.split(":")[0]) for i, l in enumerate(lines) if re.match(r"^_Z\S+:", l)]
    bad, nrot, nmfma = [], 0, 0
    for i0, name in starts:
        if only and only not in name:
            continue
        end = next((j for j in range(i0, len(lines)) if "s_endpgm" in lines[j]), len(lines))
        body = lines[i0 + 1:end + 1]
        labels = {l.strip()[:-1]: k for k, l in enumerate(body) if re.match(r"^\.LBB\d+_\d+:", l.strip())}
        for k, l in enumerate(body):
            p = parse(l)
            if not p or not p[0].startswith("v_mfma"):
                continue
            nmfma += 1
            op, ops = p
            dst, srcc = regs(ops[0]), regs(ops[3])
            if dst == srcc:
                continue
            nrot += 1
            need = min_f32 if re.search(r"x\d+_f32$", op) else min_bf16
            r = scan(body, labels, k + 1, dst, 0, need)
            if r:
                bad.append((name, op, ops[0], ops[3], r[0], r[1], need))
    return nmfma, nrot, bad


def main(argv):
    path = argv[1]
    min_f32, min_bf16, only = 16, 12, None
    a = argv[2:]
    while a:
        if a[0] == "--min-f32": min_f32 = int(a[1]); a = a[2:]
        elif a[0] == "--min-bf16": min_bf16 = int(a[1]); a = a[2:]
        elif a[0] == "--kernel": only = a[1]; a = a[2:]
        else: raise SystemExit(__doc__)
    nmfma, nrot, bad = check(path, min_f32, min_bf16, only)
    print(f"{path}: {nmfma} MFMAs, {nrot} with vDst != SrcC, {len(bad)} of them read (or left at a branch) too early")
    for b in bad[:40]:
        print("  %s\n     %s %s <- SrcC %s : %s after %d wait states (need %d)" % (b[0][-70:], b[1], b[2], b[3], b[4], b[5], b[6]))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv))
