"""Summarises rocprofv3 --pmc csv output: per kernel name, mean counter value per dispatch."""
import csv
import glob
import sys
from collections import defaultdict

root = sys.argv[1]
pat = sys.argv[2] if len(sys.argv) > 2 else "k_front"
for f in sorted(glob.glob(root + "/**/*counter_collection.csv", recursive=True)):
    acc = defaultdict(lambda: defaultdict(list))
    for row in csv.DictReader(open(f)):
        name = row.get("Kernel_Name", "")
        if pat in name:
            acc[name][row["Counter_Name"]].append(float(row["Counter_Value"]))
    for name, ctr in acc.items():
        print(f.split("/")[-3] if f.count("/") >= 3 else f, name[:60])
        for c, v in sorted(ctr.items()):
            print("    %-28s n=%3d mean=%16.1f" % (c, len(v), sum(v) / len(v)))
