#!/usr/bin/env python3
"""Instruction mix of a kernel's hottest loop in a hipcc -save-temps .s file.

    tools/loop_mix.py file.s kernel-substring [--seq]

Prints, for the longest backward-branch loop of every kernel whose mangled name contains the substring: the instruction
classes, the opcode histogram, the sizes of the VALU groups between MFMAs, and the issue cost the fp64 gap price list
(bench/ubench_gap.hip, profiles/r04_ubench_gap.txt) predicts:  a group of n VALU instructions between two fp64 MFMAs costs
about 11.3 + 4.67 n cycles of the matrix pipe, a big MFMA 64, a 4x4x4 one about 17.5, DS / SALU / s_nop next to nothing.
"""
import collections
import re
import sys


def classify(op):
    if op.startswith("v_mfma"):
        return "MFMA"
    if op.startswith("v_"):
        return "VALU"
    if op.startswith("ds_"):
        return "DS"
    if op.startswith("buffer_") or op.startswith("global_") or op.startswith("flat_") or op.startswith("scratch_"):
        return "VMEM"
    if op.startswith("s_"):
        return "SALU"
    return "other"


def main():
    path, pat = sys.argv[1], sys.argv[2]
    show_seq = "--seq" in sys.argv
    s = open(path).read()
    for m in re.finditer(r"^(\w+):\s*; @\1", s, flags=re.M):
        name = m.group(1)
        if pat not in name:
            continue
        body = s[m.end():s.index("s_endpgm", m.end())]
        lines = body.split("\n")
        labels = {}
        for k, l in enumerate(lines):
            lm = re.match(r"^(\.LBB\d+_\d+):", l)
            if lm:
                labels[lm.group(1)] = k
        loops = []
        for k, l in enumerate(lines):
            bm = re.search(r"s_c?branch\w*\s+(\.LBB\d+_\d+)", l)
            if bm and bm.group(1) in labels and labels[bm.group(1)] < k:
                loops.append((labels[bm.group(1)], k))
        if not loops:
            print(name, ": no loop")
            continue
        a, b = max(loops, key=lambda t: t[1] - t[0])
        cnt = collections.Counter()
        seq = []
        for l in lines[a:b + 1]:
            l = l.strip()
            if not l or l[0] in ";." or l.endswith(":"):
                continue
            op = l.split()[0]
            cnt[op] += 1
            seq.append(op)
        tot = collections.Counter()
        for op, c in cnt.items():
            tot[classify(op)] += c
        groups, cur = [], 0
        for op in seq:
            if op.startswith("v_mfma"):
                if cur:
                    groups.append(cur)
                cur = 0
            elif classify(op) == "VALU":
                cur += 1
        if cur:
            groups.append(cur)
        big = sum(c for op, c in cnt.items() if op.startswith("v_mfma") and "4x4x4" not in op)
        small = sum(c for op, c in cnt.items() if op.startswith("v_mfma") and "4x4x4" in op)
        valu_cost = sum(11.3 + 4.67 * g for g in groups)
        mfma_cost = 64 * big + 17.5 * small
        print(f"== {name}\n   loop lines {a}..{b}: {dict(tot)}")
        print(f"   MFMA big {big} small {small} -> {mfma_cost:.0f} cycles;  VALU {tot['VALU']} in {len(groups)} groups {sorted(groups)} -> {valu_cost:.0f} cycles"
              f"  (VALU/MFMA {tot['VALU'] / max(1, big + small):.2f}; predicted pipe busy {mfma_cost / (mfma_cost + valu_cost):.3f})")
        for op, c in sorted(cnt.items(), key=lambda t: -t[1]):
            print(f"   {c:5d} {op}")
        if show_seq:
            print("   " + " ".join(seq))


if __name__ == "__main__":
    main()
