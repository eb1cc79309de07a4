#!/usr/bin/env python3
"""Summarise rocprofv3 CSV output (kernel stats + PMC counter passes) into a small text table.
usage: summarize.py <kernel_stats.csv> [counter_collection.csv ...]"""
import csv
import re
import sys
from collections import defaultdict


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    m = re.match(r"([\w:]+(<[^(]*>)?)", name)
    s = m.group(1) if m else name
    s = s.replace("unsigned long long", "u64").replace("unsigned int", "u32").replace("bp::", "")
    return s[:60]


def main():
    stats = sys.argv[1]
    print("%-62s %6s %12s %8s" % ("kernel", "calls", "avg_us", "pct"))
    for r in csv.DictReader(open(stats)):
        print("%-62s %6s %12.1f %8s" % (short(r["Name"]), r["Calls"], float(r["AverageNs"]) / 1e3, r["Percentage"]))
    for path in sys.argv[2:]:
        acc = defaultdict(lambda: defaultdict(list))
        for r in csv.DictReader(open(path)):
            acc[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
        print("\n# %s (mean per dispatch)" % path)
        for k in sorted(acc):
            print("%-62s %s" % (k, "  ".join("%s=%.4g" % (c, sum(v) / len(v)) for c, v in sorted(acc[k].items()))))


if __name__ == "__main__":
    main()
