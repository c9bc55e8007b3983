#!/usr/bin/env python3
"""Registers, spills and LDS of every kernel in an AMDGPU .s file (hipcc -save-temps): reads the
amdhsa metadata at the end of the file.   usage: kernel_regs.py file.s [substring]"""
import re, sys
txt = open(sys.argv[1]).read()
pat = sys.argv[2] if len(sys.argv) > 2 else ""
meta = txt[txt.rindex("amdhsa.kernels:"):]
for blk in meta.split("  - .agpr_count:")[1:]:
    f = dict(re.findall(r"\.(\w+):\s+(\S+)", blk))
    name = f.get("name", "?")
    if pat in name:
        print(f"{name[:110]:110s} vgpr {f.get('vgpr_count')} agpr {blk.split()[0]} sgpr {f.get('sgpr_count')} "
              f"spill {f.get('vgpr_spill_count')} lds {f.get('group_segment_fixed_size')}")
