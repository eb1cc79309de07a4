#!/bin/bash
# tools/ab_env_frames.sh name:VAR=1,VAR2=0 ...  -> full-frame numbers (10 M step, 1 M frame, moving scene) per environment
cd "$(dirname "$0")/.."
for spec in "$@"; do
  name=${spec%%:*}; envs=${spec#*:}
  ( IFS=','; for kv in $envs; do [ -n "$kv" ] && export "$kv"; done
    python bench.py --steps 10 --warmup 3 --no-cpu-baseline ${AB_ARGS} > gpurun_out/fe_$name.json 2>/dev/null )
  python tools/show_frames.py gpurun_out/fe_$name.json
done
