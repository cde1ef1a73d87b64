"""Prints, from a rocprofv3 kernel trace (csv), the start and end of the first kernels of the library relative
to the first one: shows which launches ran side by side."""
import csv
import glob
import sys

rows = []
for p in glob.glob(sys.argv[1] + "/*kernel_trace.csv"):
    for r in csv.DictReader(open(p)):
        if "jpg::" in r["Kernel_Name"]:
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][-24:]))
rows.sort()
skip = int(sys.argv[2]) if len(sys.argv) > 2 else 40
t0 = rows[skip][0]
for s, e, n in rows[skip:skip + 24]:
    print("%9.1f %9.1f  %6.1f us  %s" % ((s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, n))
