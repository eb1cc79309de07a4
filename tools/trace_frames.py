"""Per-kernel times of the steady frames in a rocprofv3 kernel trace (a frame = encode_count .. publish).
   python tools/trace_frames.py <kernel_trace.csv> [frames_from_the_end]"""
import csv, sys, statistics as st
rows = list(csv.DictReader(open(sys.argv[1])))
tail = int(sys.argv[2]) if len(sys.argv) > 2 else 30
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows)
starts = [i for i, e in enumerate(ev) if "encode_count_kernel" in e[2]]
frames = []
for a, b in zip(starts, starts[1:]):
    seg = ev[a:b]
    pubs = [i for i, e in enumerate(seg) if "publish_kernel" in e[2]]
    if pubs:
        frames.append(seg[:pubs[-1] + 1])
n_k = st.mode(len(f) for f in frames)
frames = [f for f in frames if len(f) == n_k][-tail:]
print("%d frames of %d kernels; span median %.1f us, busy median %.1f us" % (
    len(frames), n_k, st.median(f[-1][1] - f[0][0] for f in frames) / 1e3, st.median(sum(e - s for s, e, _ in f) for f in frames) / 1e3))
for q in range(n_k):
    name = frames[0][q][2].replace("bp::(anonymous namespace)::", "").replace("void ", "").replace("bp::", "")
    print("  %7.1f us  %s" % (st.median(f[q][1] - f[q][0] for f in frames) / 1e3, name[:90]))
