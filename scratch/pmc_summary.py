"""Per-kernel averages of rocprofv3 counter_collection CSVs: pmc_summary.py <dir> <kernel substring> [tag]."""
import collections
import csv
import glob
import re
import sys

root, pat = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "k_")
tag = sys.argv[3] if len(sys.argv) > 3 else ""
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob(f"{root}/pmc_{tag}*/**/*counter_collection.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"]
        if pat in name and "at::native" not in name:
            short = re.sub(r"\(.*", "", name.replace("(anonymous namespace)::", "").replace("void ", ""))
            acc[short][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    parts = [f"{c}={sum(v) / len(v):.4g}" for c, v in sorted(d.items())]
    n = max(len(v) for v in d.values())
    print(f"{k} (n={n}): " + " ".join(parts))
