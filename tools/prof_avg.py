#!/usr/bin/env python3
"""Print per-kernel average durations from a rocprofv3 --kernel-trace CSV (optionally only kernels
whose name contains a substring, skipping the first N launches of each as warm-up)."""
import csv
import collections
import sys


def main():
    path, sub = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "")
    skip = int(sys.argv[3]) if len(sys.argv) > 3 else 0
    by = collections.OrderedDict()
    for r in csv.DictReader(open(path)):
        n = r["Kernel_Name"]
        if sub and sub not in n:
            continue
        by.setdefault(n, []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    for n, v in by.items():
        v = v[skip:] if len(v) > skip else v
        short = n.split("(")[0][-60:] if "<" not in n else n[n.index("::", 10) + 2:][:70]
        print(f"{short:70s} n={len(v):5d} avg {sum(v) / len(v):8.2f} us  min {min(v):8.2f}")


if __name__ == "__main__":
    main()
