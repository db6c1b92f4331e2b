"""Launches of one kernel (substring) between consecutive adam_kernel launches of a rocprofv3 kernel trace (csv):
python tools/copies_per_step.py <kernel_trace.csv> [substring=copyBuffer]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
sub = sys.argv[2] if len(sys.argv) > 2 else "copyBuffer"
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
seg, n, tot = [], 0, 0
for r in rows:
    if sub in r["Kernel_Name"]:
        n += 1
        tot += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    if "adam_kernel" in r["Kernel_Name"]:
        seg.append((n, tot / 1e3))
        n, tot = 0, 0
print("per step (count, us):", seg, "after the last step:", n)
