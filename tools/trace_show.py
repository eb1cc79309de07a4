"""print the last complete frame of a tools/trace_frame.py trace: every kernel / copy in start order, duration, gap to the previous end"""
import csv, glob, os, sys
d = sys.argv[1]
rows = []
for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:70]))
for f in glob.glob(os.path.join(d, "**", "*memory_copy_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY " + r.get("Direction", "") + " " + r.get("Bytes", r.get("Size", ""))))
rows.sort()
starts = [i for i, r in enumerate(rows) if "encode_count" in r[2]]
if len(starts) < 3:
    print("no frames found", len(rows)); sys.exit(1)
a, b = starts[-2], starts[-1]
t0 = rows[a][0]
prev_end = rows[a][0]
busy = 0
for s, e, name in rows[a:b]:
    print("%9.1f us  dur %7.1f  gap %6.1f  %s" % ((s - t0) / 1e3, (e - s) / 1e3, (s - prev_end) / 1e3, name))
    busy += e - s
    prev_end = max(prev_end, e)
print("frame %.1f us, busy %.1f us, ops %d" % ((rows[b][0] - t0) / 1e3, busy / 1e3, b - a))
