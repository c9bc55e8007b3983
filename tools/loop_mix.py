#!/usr/bin/env python3
"""Instruction mix of the loops of a kernel's ISA that hold at least N MFMAs.
usage: loop_mix.py kernel.s [N]"""
import re, sys
from collections import Counter
L = open(sys.argv[1]).read().split('\n')
nmin = int(sys.argv[2]) if len(sys.argv) > 2 else 8
labels = {l.split(':')[0]: i for i, l in enumerate(L) if l.startswith('.LBB')}
for i, l in enumerate(L):
    m = re.search(r's_cbranch_\w+ (\.LBB\d+_\d+)', l)
    if m and m.group(1) in labels and labels[m.group(1)] < i:
        a = labels[m.group(1)]
        ops = [x.split()[0] for x in L[a:i + 1] if x.startswith('\t') and not x.strip().startswith(';')]
        c = Counter(ops)
        nm = sum(v for k, v in c.items() if k.startswith('v_mfma'))
        if nm >= nmin and i - a < 3000:
            valu = sum(v for k, v in c.items() if k.startswith('v_') and not k.startswith('v_mfma'))
            print(f"loop lines {a}-{i}: mfma {nm} valu {valu} ({valu / (nm / 4):.2f} per step)")
            print("  ", dict(c.most_common()))
