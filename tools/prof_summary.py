"""Prints the bench value and the per-kernel averages of one rocprofv3 --kernel-trace --stats run."""
import csv
import glob
import json
import sys

log, stats_dir = sys.argv[1], sys.argv[2]
value = None
for line in open(log, errors="replace"):
    line = line.strip()
    if line.startswith("{") and '"metric"' in line:
        d = json.loads(line)
        value = "%.0f MP/s, scan %d B, front %.1f us" % (d["value"], d["config"]["scan_bytes_last_image"],
                                                        d["roofline"]["kernel_ms"] * 1e3)
print("  ", value)
for p in glob.glob(stats_dir + "/*kernel_stats.csv"):
    for row in csv.DictReader(open(p)):
        if "jpg::" in row["Name"]:
            print("   %-44s calls %5s  avg %8.1f us  min %8.1f" % (row["Name"][:44], row["Calls"],
                  float(row["AverageNs"]) / 1e3, float(row["MinNs"]) / 1e3))
