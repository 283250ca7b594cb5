#!/usr/bin/env python3
"""Issue-cost estimate of a kernel's basic blocks from a hipcc -S listing, with the per-class costs scripts/microbench/inst_cost.hip
measured on gfx950 (shader cycles a wave64 instruction occupies a SIMD with 4 resident wavefronts; profiles/r5_inst_cost.txt):
  1.62  VOP2 binary32 add / sub / mul / fmac / min / max and simple integer ops on VGPR operands, v_mov
  1.83  v_fma_f32 and other VOP3-encoded binary32 arithmetic on VGPR operands (also with |x| / -x modifiers, literals)
  2.61  VOP1 conversions / floor / trunc / fract / rndne, VOPC compares into vcc, v_mul_u32_u24, v_max/min_i32, and ANY VOP2
        with an SGPR source (v_fmac_f32 v, s, v)
  2.82  three-operand integer ops, v_med3 / v_min3 / v_max3, v_div_fixup / v_div_scale / v_div_fmas, v_ldexp, every DPP form, compares
        into an SGPR pair, v_cndmask, v_fma_f32 with an SGPR operand
  5.12  v_rcp / v_sqrt / v_rsq / v_exp / v_log, v_permlane32_swap / v_permlane16_swap
  ~1.4  per scalar ALU instruction interleaved with vector work (v_fma + one SALU each: 3.2 instead of 1.83)
usage: isa_cost.py listing.s kernel_substring [min_block_cycles]"""
import collections
import re
import sys

TRANS = re.compile(r"^v_(rcp|sqrt|rsq|exp|log|sin|cos)_|^v_permlane")
VOP3_SLOW = re.compile(r"^v_(med3|min3|max3|or3|bfe|bfi|add_lshl|lshl_add|lshl_or|and_or|add3|xad|mad_u32_u24|mad_i32_i24|mad_u64|mul_lo|mul_hi|div_|ldexp|lshlrev_b64|lshrrev_b64|ashrrev_i64|perm|alignbit|sad|cndmask|readlane|readfirstlane|writelane|mbcnt|cvt_pk)")
VOP1 = re.compile(r"^v_(cvt_|floor|trunc|fract|rndne|ceil|frexp|not_|ffb|bfrev|clz|ctz)")
FAST_INT = re.compile(r"^v_(add_u32|sub_u32|subrev_u32|and_b32|or_b32|xor_b32|lshlrev_b32|lshrrev_b32|ashrrev_i32|mov_b32|add_co|sub_co|addc_co|subb_co|mov_b64)")
FAST_FP = re.compile(r"^v_(add|sub|subrev|mul|fmac|max|min|mac)_(f32|legacy_f32)")
FMA = re.compile(r"^v_(fma|fmaak|fmamk|mad)_f32")


def cost(line):
    parts = line.replace(",", " ").split()
    op = parts[0]
    has_sgpr = any(re.match(r"^-?\|?s\d+|^-?\|?s\[", a) for a in parts[1:]) or " vcc" in line and op.startswith("v_cndmask")
    if not op.startswith("v_"):
        if op.startswith("s_nop"):
            return "nop", 0.4
        if op.startswith("s_waitcnt"):
            return "wait", 0.0
        if op.startswith("s_"):
            return "salu", 1.4
        return "mem", 1.0
    if "dpp" in line or "sdwa" in line:
        return "dpp/sdwa", 2.82
    if TRANS.match(op):
        return "trans/permlane", 5.12
    if op.startswith("v_cmp"):
        return "cmp", 2.87 if "_e64" in op else 2.68
    if VOP3_SLOW.match(op):
        return "vop3-int/select", 2.82
    if VOP1.match(op):
        return "vop1", 2.61
    if re.match(r"^v_(mul_u32_u24|mul_i32_i24|max_i32|min_i32|max_u32|min_u32)", op):
        return "vop2-slow-int", 2.61
    if FMA.match(op):
        return ("fma+sgpr", 2.82) if has_sgpr else ("fma", 1.83)
    if FAST_FP.match(op):
        if "_e64" in op:
            return ("fp-vop3+sgpr", 2.82) if has_sgpr else ("fp-vop3", 1.83)
        return ("fp+sgpr", 2.61) if has_sgpr else ("fp", 1.62)
    if FAST_INT.match(op):
        return ("int+sgpr", 2.61) if has_sgpr else ("int", 1.62)
    return "other-valu", 2.82


def main():
    text = open(sys.argv[1]).read().split("\n")
    start = next(i for i, l in enumerate(text) if re.match(r"^_Z\w*:", l) and sys.argv[2] in l)
    min_cycles = float(sys.argv[3]) if len(sys.argv) > 3 else 40.0
    blocks, cur = collections.OrderedDict(), "entry"
    blocks[cur] = []
    depth = {cur: 0}
    for l in text[start + 1:]:
        if re.match(r"^\.Lfunc_end", l):
            break
        m = re.match(r"^(\.LBB\w+):(.*)", l)
        if m:
            cur = m.group(1)
            blocks[cur] = []
            d = re.search(r"Depth=(\d+)", m.group(2))
            depth[cur] = int(d.group(1)) if d else 0
            continue
        d = re.search(r";\s+(?:in Loop: Header=\w+|=>\s+This Inner Loop Header:) Depth=(\d+)", l)
        if l.startswith("\t") and not l.strip().startswith((".", ";")):
            blocks[cur].append(l.strip())
    grand = collections.Counter()
    for name, ins in blocks.items():
        c, n = collections.Counter(), collections.Counter()
        for i in ins:
            k, v = cost(i)
            c[k] += v
            n[k] += 1
        total = sum(c.values())
        for k in c:
            grand[(depth.get(name, 0), k)] += c[k]
        if total >= min_cycles:
            print(f"{name:12s} depth {depth.get(name, 0)} {total:7.1f} cycles  " + "  ".join(f"{k}={n[k]}/{c[k]:.0f}" for k in sorted(c, key=lambda k: -c[k]) if c[k] >= 1))
    deepest = max(d for d, _ in grand)
    tot = sum(v for (d, k), v in grand.items() if d == deepest)
    print(f"innermost loop (depth {deepest}), every block counted once: {tot:.0f} cycles: " +
          "  ".join(f"{k}={v:.0f}" for (d, k), v in sorted(grand.items(), key=lambda kv: -kv[1]) if d == deepest))


if __name__ == "__main__":
    main()
