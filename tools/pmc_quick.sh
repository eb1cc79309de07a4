#!/bin/bash
# tools/pmc_quick.sh <tag> [counter ...]  -> one rocprofv3 PMC pass over a short 10 M bench run, mean counters per kernel
TAG=$1; shift
CTRS=${@:-SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES}
R=$(pwd); OUT=$R/gpurun_out/pmc_$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --pmc $CTRS --kernel-trace --output-format csv -d $OUT -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-1m --no-moving --no-extra > $OUT/log.txt 2>&1
rc=$?
cd $R
python3 - <<PY
import csv, glob, collections
f = glob.glob("$OUT/**/*counter_collection.csv", recursive=True)
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for p in f:
    for r in csv.DictReader(open(p)):
        acc[r["Kernel_Name"][:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(acc):
    if any(s in k for s in ("encode", "finish", "scan_expand_kernel", "radix_claim")):
        print("%-72s %s" % (k, "  ".join("%s=%.3g" % (c.replace("SQ_", ""), sum(v) / len(v)) for c, v in sorted(acc[k].items()))))
PY
find $OUT -name '*.csv' -size +2M -delete
exit $rc
