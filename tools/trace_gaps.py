"""Reads a rocprofv3 kernel-trace csv and reports, over the steady tail of the run, how much of the stream's time is kernels and how
much is gaps between them (launch latency, host work).   python tools/trace_gaps.py <kernel_trace.csv> [tail_fraction]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
frac = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows)
ev = ev[int(len(ev) * (1 - frac)):]
busy = sum(e - s for s, e, _ in ev)
span = ev[-1][1] - ev[0][0]
gaps = [ev[i + 1][0] - ev[i][1] for i in range(len(ev) - 1)]
gaps_pos = [g for g in gaps if g > 0]
print("kernels %d  span %.3f ms  busy %.3f ms (%.1f %%)  gaps: n %d  mean %.2f us  median %.2f us  sum %.3f ms" % (
    len(ev), span / 1e6, busy / 1e6, 100.0 * busy / span, len(gaps_pos), sum(gaps_pos) / max(1, len(gaps_pos)) / 1e3,
    sorted(gaps_pos)[len(gaps_pos) // 2] / 1e3 if gaps_pos else 0, sum(gaps_pos) / 1e6))
big = sorted(((g, ev[i][2][:50], ev[i + 1][2][:50]) for i, g in enumerate(gaps)), reverse=True)[:8]
for g, a, b in big:
    print("  gap %.1f us after %s before %s" % (g / 1e3, a, b))
from collections import defaultdict
after = defaultdict(list)
for i, g in enumerate(gaps):
    after[ev[i][2][:60]].append(g)
print("mean gap AFTER each kernel (us), by total:")
for k, v in sorted(after.items(), key=lambda kv: -sum(kv[1]))[:24]:
    print("  %-60s n %4d  mean %6.2f  total %.3f ms" % (k, len(v), sum(v) / len(v) / 1e3, sum(v) / 1e6))
