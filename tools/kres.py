"""Kernel resource usage from a hipcc -S file: python tools/kres.py file.s"""
import re, sys
t = open(sys.argv[1]).read()
for blk in t.split("  - .agpr_count:")[1:]:
    g = lambda k: (re.search(r"\." + k + r":\s+(\S+)", blk) or [None, "?"])[1]
    name = g("name")
    print(f"{name[:60]:60s} vgpr={g('vgpr_count'):>4s} agpr={blk.split()[0]:>3s} sgpr={g('sgpr_count'):>4s} lds={g('group_segment_fixed_size'):>6s} scratch={g('private_segment_fixed_size'):>5s} spill={g('vgpr_spill_count')}")
