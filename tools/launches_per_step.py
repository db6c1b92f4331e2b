"""Kernel launches between consecutive adam_kernel launches of a rocprofv3 kernel trace (csv) = the nodes of one replayed step graph
(the per-step tables of profiles/ divide a whole command's launches by its steps, so one-time work — parameter flattening copies, the
plan-discovery forward's packs — shows up as a fraction per step there).   python tools/launches_per_step.py <kernel_trace.csv>"""
import collections, csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
segs, cur = [], collections.Counter()
for r in rows:
    name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")
    cur[name.split("(")[0][:70]] += 1
    if "adam_kernel" in r["Kernel_Name"]:
        segs.append(cur)
        cur = collections.Counter()
print("launches per step:", [sum(s.values()) for s in segs])
if len(segs) >= 2:
    last = segs[-1]
    print("the last step by kernel:")
    for k, v in sorted(last.items(), key=lambda kv: -kv[1]):
        print(f"{v:5d}  {k}")
