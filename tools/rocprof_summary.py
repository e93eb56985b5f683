#!/usr/bin/env python
"""Condenses rocprofv3 CSV output into the small per-kernel tables committed
under profiles/: average duration per kernel from *_kernel_trace.csv and the
per-kernel mean of every PMC counter from *_counter_collection.csv.

  python tools/rocprof_summary.py <rocprof output dir> > profiles/<name>.md
"""
import csv
import glob
import os
import sys
from collections import defaultdict


def short(name):
    for p in ("void ", "vx::(anonymous namespace)::", "(anonymous namespace)::", "vx::"):
        name = name.replace(p, "")
    name = name.split("(")[0]
    return name[:60]


def main(root):
    traces = glob.glob(os.path.join(root, "**", "*kernel_trace.csv"), recursive=True)
    counters = glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True)
    for path in traces:
        dur = defaultdict(list)
        with open(path) as f:
            for row in csv.DictReader(f):
                dur[short(row["Kernel_Name"])].append(int(row["End_Timestamp"]) - int(row["Start_Timestamp"]))
        total = sum(sum(v) for v in dur.values())
        print(f"## kernel trace: {os.path.relpath(path, root)}\n")
        print("| kernel | calls | total ms | avg us | min us | max us | % |")
        print("|---|---:|---:|---:|---:|---:|---:|")
        for k, v in sorted(dur.items(), key=lambda kv: -sum(kv[1])):
            print(f"| {k} | {len(v)} | {sum(v) / 1e6:.3f} | {sum(v) / len(v) / 1e3:.2f} | "
                  f"{min(v) / 1e3:.2f} | {max(v) / 1e3:.2f} | {100 * sum(v) / total:.1f} |")
        print()
    for path in counters:
        agg = defaultdict(lambda: defaultdict(list))
        with open(path) as f:
            for row in csv.DictReader(f):
                agg[short(row["Kernel_Name"])][row["Counter_Name"]].append(float(row["Counter_Value"]))
        print(f"## counters (mean per dispatch): {os.path.relpath(path, root)}\n")
        names = sorted({c for k in agg.values() for c in k})
        print("| kernel | dispatches | " + " | ".join(names) + " |")
        print("|---|---:|" + "---:|" * len(names))
        for k, cs in sorted(agg.items()):
            n = max(len(v) for v in cs.values())
            cells = [f"{sum(cs[c]) / len(cs[c]):.4g}" if c in cs else "" for c in names]
            print(f"| {k} | {n} | " + " | ".join(cells) + " |")
        print()


if __name__ == "__main__":
    main(sys.argv[1])
