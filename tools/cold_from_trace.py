#!/usr/bin/env python3
"""From a rocprofv3 --kernel-trace CSV of `python3 bench.py` (roofline on): separate the COLD REPLAY launches of an op — the
runs of >= `minrun` consecutive dispatches of kernels whose name contains `sub`, with nothing else in between: that is how
bench.py replays the recorded launches of one C-ABI entry point back to back — from its launches inside training steps, and
print the average duration of both populations (per kernel variant).
usage: cold_from_trace.py <kernel_trace.csv> <substring> [minrun=40]"""
import collections
import csv
import sys

path, sub = sys.argv[1], sys.argv[2]
minrun = int(sys.argv[3]) if len(sys.argv) > 3 else 40
rows = sorted(csv.DictReader(open(path)), key=lambda r: int(r["Start_Timestamp"]))
runs, cur = [], []
for r in rows:
    if sub in r["Kernel_Name"]:
        cur.append(r)
    else:
        if cur:
            runs.append(cur)
        cur = []
if cur:
    runs.append(cur)
cold, step = collections.defaultdict(list), collections.defaultdict(list)
for run in runs:
    dst = cold if len(run) >= minrun else step
    for r in run:
        name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
        dst[name].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for title, d in (("cold replay (runs of >= %d back-to-back launches)" % minrun, cold), ("inside training steps / forward graphs", step)):
    allv = [v for vs in d.values() for v in vs]
    if not allv:
        continue
    print(f"{title}: {len(allv)} launches, mean {sum(allv) / len(allv):.2f} us")
    for k in sorted(d):
        v = d[k]
        print(f"   {k:60s} n={len(v):5d} mean {sum(v) / len(v):8.2f} us  min {min(v):8.2f}")
