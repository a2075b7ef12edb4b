"""Sum the counters of a rocprofv3 --pmc CSV per kernel name (substring filter optional), averaged per launch.
usage: python tools/pmc_by_kernel.py <dir> [substring]"""
import csv
import glob
import sys
from collections import defaultdict

d, sub = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "")
acc, launches = defaultdict(lambda: defaultdict(float)), defaultdict(set)
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][:70]
        if sub in k:
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
            launches[k].add(r["Dispatch_Id"])
for k in sorted(acc):
    n = len(launches[k])
    print(k, "launches", n)
    for c, v in sorted(acc[k].items()):
        print(f"   {c:32s} {v / n:16.0f}")
