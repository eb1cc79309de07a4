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
    traffic = defaultdict(dict)
    for path in sys.argv[2:]:
        acc = defaultdict(lambda: defaultdict(list))
        for r in csv.DictReader(open(path)):
            acc[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
        print("\n# %s (mean per dispatch)" % path)
        for k in sorted(acc):
            print("%-62s %s" % (k, "  ".join("%s=%.4g" % (c, sum(v) / len(v)) for c, v in sorted(acc[k].items()))))
            for c, v in acc[k].items():
                if c in ("FETCH_SIZE", "WRITE_SIZE"):
                    traffic[k][c] = sum(v) / len(v)
    # HBM bytes per launch: FETCH_SIZE / WRITE_SIZE are KiB; on gfx950 FETCH_SIZE counts half of a
    # coalesced >= 8 B/lane read stream (MI355X_MICROARCH.md, HBM section), so it is doubled
    out = {k: (2.0 * t.get("FETCH_SIZE", 0.0) + t.get("WRITE_SIZE", 0.0)) * 1024.0 for k, t in traffic.items()
           if "FETCH_SIZE" in t and "WRITE_SIZE" in t}
    if out:
        print("\n# HBM bytes per launch = (2 x FETCH_SIZE + WRITE_SIZE) KiB")
        for k in sorted(out):
            print("%-62s %.4g" % (k, out[k]))
        import json
        print("TRAFFIC_JSON " + json.dumps(out))


if __name__ == "__main__":
    main()
