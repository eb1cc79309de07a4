#!/bin/bash
# tools/kt_quick.sh <tag> [ENV=1 ...] -> kernel-trace stats of a short 10 M bench run (top kernels by time), env switches optional
TAG=$1; shift
R=$(pwd); OUT=$R/gpurun_out/kt_$TAG; mkdir -p $OUT
for kv in "$@"; do export "$kv"; done
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-1m --no-moving --no-extra > $OUT/log.txt 2>&1
rc=$?
cd $R
KS=$(find $OUT -name '*kernel_stats.csv' | head -1)
cp $KS gpurun_out/kt_${TAG}_kernel_stats.csv
python3 - <<EOF2
import csv
rows=list(csv.DictReader(open("gpurun_out/kt_${TAG}_kernel_stats.csv")))
rows.sort(key=lambda r:-float(r["TotalDurationNs"]))
print("== $TAG")
for r in rows[:14]:
    print("%9.1f us x%-4s %s" % (float(r["AverageNs"])/1e3, r["Calls"], r["Name"][:110]))
EOF2
find $OUT -name '*.csv' -size +2M -delete
exit $rc
