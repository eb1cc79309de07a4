#!/bin/bash
# A/B of run-time switches in ONE gpurun call: tools/ab_env.sh name1:VAR=1,VAR2=0 name2: ...   (bench.py per variant
# -> gpurun_out/ab_<name>.json; AB_ARGS adds bench arguments, AB_LIB=<variant> picks a library built by tools/ab.sh build)
cd "$(dirname "$0")/.."
for spec in "$@"; do
  name=${spec%%:*}; envs=${spec#*:}
  lib=broadphase_rs_amd/lib/libbroadphase_amd.so
  [ -n "$AB_LIB" ] && lib=broadphase_rs_amd/lib_ab/$AB_LIB/libbroadphase_amd.so
  (
    IFS=','; for kv in $envs; do [ -n "$kv" ] && export "$kv"; done
    BP_LIBRARY=$PWD/$lib timeout -k 10 240 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-1m --no-moving --no-extra ${AB_ARGS} > gpurun_out/ab_$name.json 2> gpurun_out/ab_$name.err
  ) || { echo "variant $name failed (rc $?)"; tail -3 gpurun_out/ab_$name.err; }
  python tools/ab_show.py $name
done
