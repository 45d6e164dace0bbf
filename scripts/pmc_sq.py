"""Summarise rocprofv3 --pmc csv passes: per kernel-name fragment, average of every counter per launch.
    python scripts/pmc_sq.py <fragment> dir [dir ...]"""
import csv
import glob
import os
import sys

frag = sys.argv[1]
agg = {}
for d in sys.argv[2:]:
    for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(path, newline="") as fh:
            for row in csv.DictReader(fh):
                if frag not in row["Kernel_Name"]:
                    continue
                a = agg.setdefault(row["Counter_Name"], [0, 0.0])
                a[0] += 1
                a[1] += float(row["Counter_Value"])
for k in sorted(agg):
    n, v = agg[k]
    print("%-36s launches %4d   avg %16.1f" % (k, n, v / n))
