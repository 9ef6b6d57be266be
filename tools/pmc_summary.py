"""Summarise rocprofv3 --pmc CSVs for one kernel (mean per dispatch).  Usage: pmc_summary.py [dir] [kernel name part]"""
import csv, glob, os, sys, collections
root = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/pmc"
kernel = sys.argv[2] if len(sys.argv) > 2 else "mjrl_step_kernel"
acc = collections.defaultdict(list)
for path in glob.glob(os.path.join(root, "*", "*", "*counter_collection.csv")):
    for row in csv.DictReader(open(path)):
        if kernel in row.get("Kernel_Name", ""):
            acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
for k in sorted(acc):
    v = acc[k]
    print(f"{k:32s} mean/dispatch {sum(v)/len(v):16.1f}  (n={len(v)})")
