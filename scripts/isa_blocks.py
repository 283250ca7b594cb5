#!/usr/bin/env python3
"""Per-basic-block instruction classes of one kernel in a hipcc -S listing: where the VALU instructions of a sweep go
(arithmetic vs moves / selects / compares / cross-lane / conversions).
usage: isa_blocks.py listing.s kernel_substring [min_block_size]"""
import collections
import re
import sys

CLASSES = [
    ("fp_arith", r"^v_(fma|fmac|mul|add|sub|mac|mad|rcp|rsq|sqrt|exp|log|max|min|med3|ldexp|frexp|fract|floor|trunc|rndne|ceil|div_|pk_).*_(f32|f64|f16|legacy_f32)"),
    ("cvt", r"^v_cvt"),
    ("cmp", r"^v_cmp"),
    ("select", r"^v_cndmask"),
    ("mov", r"^v_(mov|accvgpr|swap)"),
    ("lane", r"^v_(readlane|readfirstlane|writelane|permlane|bpermute|mov_b32_dpp)|_dpp|ds_swizzle|ds_bpermute"),
    ("int_arith", r"^v_(add|sub|mul|mad|lshl|lshr|ashr|and|or|xor|not|bfe|bfi|alignbit|min|max|med3|add3|lshl_add|lshl_or|and_or|or3|xad|perm|ffb|mbcnt|sad).*"),
    ("valu_other", r"^v_"),
    ("vmem", r"^(global|flat|buffer|scratch)_"),
    ("lds", r"^ds_"),
    ("smem", r"^s_(load|buffer_load)"),
    ("wait", r"^s_waitcnt"),
    ("branch", r"^s_(cbranch|branch)"),
    ("salu", r"^s_"),
]


def classify(op, line):
    if "dpp" in line and op.startswith("v_") and not op.startswith("v_mov"):
        return "fp_arith" if re.search(r"_f32|_f64", op) else "int_arith"
    for name, pat in CLASSES:
        if re.search(pat, op):
            return name
    return "other"


def main():
    text = open(sys.argv[1]).read().split("\n")
    start = next(i for i, l in enumerate(text) if re.match(r"^_Z\w*:", l) and sys.argv[2] in l)
    min_size = int(sys.argv[3]) if len(sys.argv) > 3 else 40
    blocks, cur = collections.OrderedDict(), "entry"
    blocks[cur] = []
    for l in text[start + 1:]:
        if re.match(r"^\.Lfunc_end", l):
            break
        m = re.match(r"^(\.LBB\w+):", l)
        if m:
            cur = m.group(1)
            blocks[cur] = []
            continue
        if l.startswith("\t") and not l.strip().startswith((".", ";")):
            blocks[cur].append(l.strip())
    total = collections.Counter()
    for name, ins in blocks.items():
        c = collections.Counter(classify(i.split()[0], i) for i in ins)
        total.update(c)
        if len(ins) >= min_size:
            valu = sum(v for k, v in c.items() if k in ("fp_arith", "cvt", "cmp", "select", "mov", "lane", "int_arith", "valu_other"))
            print(f"{name:14s} n={len(ins):4d} valu={valu:4d} " + " ".join(f"{k}={c[k]}" for k in ("fp_arith", "int_arith", "cvt", "cmp", "select", "mov", "lane", "valu_other", "vmem", "lds", "smem", "wait", "salu", "branch") if c[k]))
    valu = sum(v for k, v in total.items() if k in ("fp_arith", "cvt", "cmp", "select", "mov", "lane", "int_arith", "valu_other"))
    print("TOTAL", sum(total.values()), "valu", valu, dict(total))


if __name__ == "__main__":
    main()
