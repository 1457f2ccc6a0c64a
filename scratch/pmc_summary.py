"""Sum rocprofv3 counter_collection CSVs per kernel name prefix (average per dispatch)."""
import csv, sys, glob, collections
pat = sys.argv[2] if len(sys.argv) > 2 else "k_gen_bwd"
for f in sorted(glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if pat in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        print(f"{k:32s} n={len(v):3d} avg={sum(v)/len(v):.4g}")
