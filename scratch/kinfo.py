"""Registers, spills and LDS of the kernels of a hipcc -S listing whose names contain a substring: kinfo.py file.s [substr]"""
import re, sys
txt = open(sys.argv[1]).read()
key = sys.argv[2] if len(sys.argv) > 2 else ""
for blk in txt.split("  - .agpr_count:")[1:]:
    name = re.search(r"\.name:\s+(\S+)", blk)
    if not name or key not in name.group(1): continue
    g = lambda k: (re.search(r"\.%s:\s+(\d+)" % k, blk) or [None, "?"])[1]
    n = re.sub(r"^_ZN12_GLOBAL__N_1\d+", "", name.group(1))[:44]
    print("%-46s vgpr %s spill %s sgpr %s lds %s" % (n, g("vgpr_count"), g("vgpr_spill_count"), g("sgpr_count"), g("group_segment_fixed_size")))
