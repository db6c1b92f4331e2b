#!/usr/bin/env python3
"""Print a rocprofv3 --stats kernel_stats.csv compactly: name, calls, avg us, total ms."""
import csv, sys, re
rows = list(csv.DictReader(open(sys.argv[1])))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 40
for r in rows[:n]:
    name = re.sub(r"\(anonymous namespace\)::", "", r["Name"])
    name = re.sub(r"^void ", "", name).split("(")[0]
    print(f"{name[:90]:90s} n={int(r['Calls']):6d} avg={float(r['AverageNs'])/1e3:9.1f}us tot={float(r['TotalDurationNs'])/1e6:9.2f}ms {float(r['Percentage']):5.1f}%")
