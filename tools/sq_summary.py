"""Per-kernel means of the SQ counters collected by tools/sq_profile.sh -> <dir>/sq_summary.txt"""
import collections
import csv
import glob
import os
import sys

d = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for p in glob.glob(os.path.join(d, "sq*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(p)):
        agg[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
cols = ["SQ_WAVES", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_INSTS_LDS",
        "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_VMEM", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY",
        "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE"]
with open(os.path.join(d, "sq_summary.txt"), "w") as f:
    f.write("# mean per launch; SQ_WAVE_CYCLES / WAIT / ACTIVE count quad-cycles summed over waves\n")
    f.write("kernel," + ",".join(cols) + ",launches\n")
    for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1].get("SQ_WAVE_CYCLES", [0]))):
        if not (k.startswith("k_") or k.startswith("void k_")):
            continue
        n = max(len(x) for x in v.values())
        f.write(k[:40].replace(",", ";") + "," + ",".join(f"{sum(v[c]) / len(v[c]):.4g}" if c in v else "" for c in cols) + f",{n}\n")
print(open(os.path.join(d, "sq_summary.txt")).read())
