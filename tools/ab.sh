#!/bin/bash
# A/B builds of the HIP library with different compile-time knobs, measured in ONE gpurun call:
#   tools/ab.sh build name "-DSX_BLOCK_DEF=128"      (here: cross-compiles into broadphase_rs_amd/lib_ab/name)
#   tools/ab.sh run name1 name2 ...                  (on the GPU box: bench.py per variant -> gpurun_out/ab_<name>.json)
set -e
cd "$(dirname "$0")/.."
case "$1" in
  build) make -C broadphase_rs_amd/csrc -s OUT=../lib_ab/$2 EXTRA="$3" all ;;
  run) shift; for v in "$@"; do
         lib=broadphase_rs_amd/lib_ab/$v/libbroadphase_amd.so; [ "$v" = base ] && lib=broadphase_rs_amd/lib/libbroadphase_amd.so
         BP_LIBRARY=$PWD/$lib python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-1m --no-moving --no-extra ${AB_ARGS} > gpurun_out/ab_$v.json 2> gpurun_out/ab_$v.err || echo "variant $v failed"
         python tools/ab_show.py $v
       done ;;
esac
