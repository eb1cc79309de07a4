#!/bin/bash
# tools/ab_frames.sh v1 v2 ...  -> full-frame numbers (10 M step, 1 M frame, moving scene) per library variant (base = lib/)
cd "$(dirname "$0")/.."
for v in "$@"; do
  lib=broadphase_rs_amd/lib_ab/$v/libbroadphase_amd.so; [ "$v" = base ] && lib=broadphase_rs_amd/lib/libbroadphase_amd.so
  BP_LIBRARY=$PWD/$lib python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/fr_$v.json 2>/dev/null
  python tools/show_frames.py gpurun_out/fr_$v.json
done
