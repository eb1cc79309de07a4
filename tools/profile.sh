#!/bin/bash
# Profiles `bench.py` on the GPU box: kernel-trace stats + separate PMC passes (FETCH_SIZE, WRITE_SIZE,
# SQ instruction mix), then writes gpurun_out/<tag>_summary.txt via profiles/summarize.py.
# usage (through gpurun, from the repo root):  bash tools/profile.sh <tag> [bench args...]
set -o pipefail
TAG=${1:-prof}; shift
R=$(pwd)
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 3 --warmup 1 --no-cpu-baseline --no-1m --no-moving --no-extra $*"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -- python3 $R/bench.py $ARGS > $OUT/kt.log 2>&1 &&
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch -- python3 $R/bench.py $ARGS > $OUT/fetch.log 2>&1 &&
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/write -- python3 $R/bench.py $ARGS > $OUT/write.log 2>&1 &&
timeout -k 10 300 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT --kernel-trace --output-format csv -d $OUT/sq -- python3 $R/bench.py $ARGS > $OUT/sq.log 2>&1
echo "rc=$?"
# second SQ pass (where the waves' cycles go); a counter this ROCm does not know only loses this pass
timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM --kernel-trace --output-format csv -d $OUT/sq2 -- python3 $R/bench.py $ARGS > $OUT/sq2.log 2>&1 || echo "sq2 pass failed (see sq2.log)"
cd $R
KS=$(find $OUT/kt -name '*kernel_stats.csv' | head -1)
CC=$(find $OUT/fetch $OUT/write $OUT/sq $OUT/sq2 -name '*counter_collection.csv' | sort)
python3 profiles/summarize.py $KS $CC > $R/gpurun_out/${TAG}_summary.txt 2>&1
cp $KS $R/gpurun_out/${TAG}_kernel_stats.csv
find $OUT -name '*.csv' -size +2M -delete   # keep the merge-back small
head -30 $R/gpurun_out/${TAG}_summary.txt
