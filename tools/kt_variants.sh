#!/bin/bash
# tools/kt_variants.sh v1 v2 ...  -> rocprofv3 kernel-trace of a short 10 M bench run per library variant (broadphase_rs_amd/lib_ab/<v>;
# "base" = the tree's library): the scan / finish kernels' mean durations.  For timing experiments whose RESULTS may be wrong.
R=$(pwd)
for v in "$@"; do
  lib=$R/broadphase_rs_amd/lib_ab/$v/libbroadphase_amd.so; [ "$v" = base ] && lib=$R/broadphase_rs_amd/lib/libbroadphase_amd.so
  OUT=$R/gpurun_out/ktv_$v; mkdir -p $OUT
  ( cd /tmp && export TMPDIR=/tmp && BP_LIBRARY=$lib timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-1m --no-moving --no-extra > $OUT/log.txt 2>&1 )
  KS=$(find $OUT -name '*kernel_stats.csv' | head -1)
  python3 - <<PY
import csv
rows=list(csv.DictReader(open("$KS")))
out=[]
for r in rows:
    n=r["Name"]
    for k in ("scan_expand_kernel","tgroup_finishc","group_finish_kernel<true>","radix_claim_kernel<bp::(anonymous namespace)::NoVal, false, 0>"):
        if k in n: out.append("%s %.1f x%s" % (k[:22], float(r["AverageNs"])/1e3, r["Calls"]))
print("$v:", " | ".join(out))
PY
  find $OUT -name '*.csv' -size +1M -delete
done
