#!/bin/bash
# tools/frame_timeline.sh <objects> -> one steady-state frame's kernels with the idle gap before each (rocprofv3 kernel trace)
N=${1:-1000000}
R=$(pwd); OUT=$R/gpurun_out/tl_$N; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $OUT -- python3 $R/bench.py --objects $N --steps 20 --warmup 5 --no-cpu-baseline --no-1m --no-moving --no-extra > $OUT/log.txt 2>&1
rc=$?
cd $R
python3 - <<EOF2
import csv, glob
kt = glob.glob("$OUT/**/*kernel_trace.csv", recursive=True)[0]
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(kt))]
mc = glob.glob("$OUT/**/*memory_copy_trace.csv", recursive=True)
for f in mc:
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "MEMCPY " + r.get("Direction", "")))
rows.sort()
# frames: split at encode_count launches; take the one 8 frames before the last encode_count (inside the timed region)
idx = [i for i, r in enumerate(rows) if "encode_count" in r[2]]
a, b = idx[-9], idx[-8]
fr = rows[a:b + 1]
busy = 0
print("frame of %d kernels, %.1f us from first start to next frame's first start" % (b - a, (rows[b][0] - rows[a][0]) / 1e3))
prev_end = None
for s, e, n in fr[:-1]:
    gap = (s - prev_end) / 1e3 if prev_end else 0.0
    busy += (e - s)
    short = n.replace("void ", "").replace("bp::(anonymous namespace)::", "").replace("bp::", "").split("(")[0][:60]
    print("  gap %6.1f  run %6.1f  %s" % (gap, (e - s) / 1e3, short))
    prev_end = max(prev_end or 0, e)
print("  gap %6.1f  (to the next frame's first kernel)" % ((rows[b][0] - prev_end) / 1e3))
print("busy %.1f us" % (busy / 1e3))
EOF2
find $OUT -name '*.csv' -size +2M -delete
exit $rc
