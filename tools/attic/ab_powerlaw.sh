#!/bin/bash
# tools/ab_powerlaw.sh v1 v2 ... -> mixed-scale scene (tools/bench_powerlaw.py) per library variant
cd "$(dirname "$0")/.."
for v in "$@"; do
  lib=broadphase_rs_amd/lib_ab/$v/libbroadphase_amd.so; [ "$v" = base ] && lib=broadphase_rs_amd/lib/libbroadphase_amd.so
  echo "== $v"; BP_LIBRARY=$PWD/$lib python tools/bench_powerlaw.py --objects ${PL_OBJECTS:-12500000} --steps 6 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],3), d['pairs'], {k:v for k,v in d['kernel_ms'].items() if v>0.05})"
done
