"""One frame of a rocprofv3 run as a timeline -- every kernel AND memory copy with the idle time in front of it (what is between and
beside the kernels: host round trips, allocations, small copies, kernels nobody needed).
   rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d DIR -- python3 <script>
   python tools/trace_timeline.py DIR [anchor kernel substring = encode_count_kernel] [frame index from the end = 3]"""
import csv, glob, sys
d = sys.argv[1]
anchor = sys.argv[2] if len(sys.argv) > 2 else "encode_count_kernel"
back = int(sys.argv[3]) if len(sys.argv) > 3 else 3
ev = []
for f in glob.glob(d + "/*/*kernel_trace.csv") + glob.glob(d + "/*kernel_trace.csv"):
    ev += [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f))]
for f in glob.glob(d + "/*/*memory_copy_trace.csv") + glob.glob(d + "/*memory_copy_trace.csv"):
    ev += [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY " + r.get("Direction", "")) for r in csv.DictReader(open(f))]
ev.sort()
starts = [i for i, e in enumerate(ev) if anchor in e[2]]
if len(starts) < back + 1:
    sys.exit("anchor %r found %d times" % (anchor, len(starts)))
a, b = starts[-back - 1], starts[-back]
seg = ev[a:b]
t0, prev, idle = seg[0][0], None, 0.0
for s, e, n in seg:
    gap = (s - prev) / 1e3 if prev else 0.0
    idle += max(gap, 0.0)
    print("%9.1f  gap %7.1f  dur %8.1f  %s" % ((s - t0) / 1e3, gap, (e - s) / 1e3, n.replace("bp::(anonymous namespace)::", "").replace("void ", "")[:72]))
    prev = e
print("%d events, span %.1f us, idle inside %.1f us, period %.1f us" % (len(seg), (seg[-1][1] - t0) / 1e3, idle, (ev[b][0] - t0) / 1e3))
