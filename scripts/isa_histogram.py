#!/usr/bin/env python3
"""Static instruction histogram of one kernel in a hipcc -S listing (scratch tool for the VALU-bound kernels).
usage: isa_histogram.py listing.s kernel_substring"""
import collections
import re
import sys

text = open(sys.argv[1]).read().split("\n")
start = next(i for i, l in enumerate(text) if re.match(r"^_Z\w*:", l) and sys.argv[2] in l)
hist = collections.Counter()
for l in text[start + 1:]:
    if l.strip().startswith("s_endpgm"):
        break
    if l.startswith("\t") and not l.strip().startswith((".", ";")):
        hist[l.split()[0]] += 1
total = sum(hist.values())
valu = sum(c for k, c in hist.items() if k.startswith("v_"))
print(f"total {total}  valu {valu}  salu {sum(c for k, c in hist.items() if k.startswith('s_'))}")
for k, c in hist.most_common(25):
    print(f"  {k:28s} {c}")
