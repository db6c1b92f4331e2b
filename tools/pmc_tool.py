#!/usr/bin/env python3
"""Hardware counters of ONE tool command, per kernel: python3 tools/pmc_tool.py "<CTR1 CTR2 ...>" <kernel-substring> -- python3 tools/x.py args
(one rocprofv3 --kernel-trace --pmc pass; run on the GPU box)."""
import csv, glob, os, subprocess, sys, collections
i = sys.argv.index("--")
ctrs, sub, cmd = sys.argv[1].split(), sys.argv[2], sys.argv[i + 1:]
out = "/tmp/pmc_tool_%d" % os.getpid()
r = subprocess.run(["rocprofv3", "--kernel-trace", "--pmc", *ctrs, "--output-format", "csv", "-d", out, "-o", "p", "--", *cmd],
                   env=dict(os.environ, TMPDIR="/tmp"), capture_output=True, text=True)
if r.returncode:
    print(r.stderr[-1500:]); raise SystemExit(r.returncode)
f = glob.glob(out + "/**/*counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for row in csv.DictReader(open(f)):
    k = row["Kernel_Name"]
    if sub in k:
        kk = k.replace("(anonymous namespace)::", "").replace("void ", "")
        acc[kk.split("(")[0][:70]][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, d in acc.items():
    print(k)
    for c, v in d.items():
        print(f"   {c:32s} n={len(v):4d} mean {sum(v)/len(v):16.1f}")
