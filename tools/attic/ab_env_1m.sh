#!/bin/bash
# tools/ab_env_1m.sh name:VAR=1,... ...  -> the 1 M frame (bench.py --objects 1000000) per environment
cd "$(dirname "$0")/.."
for spec in "$@"; do
  name=${spec%%:*}; envs=${spec#*:}
  ( IFS=','; for kv in $envs; do [ -n "$kv" ] && export "$kv"; done
    python bench.py --objects 1000000 --steps 60 --warmup 10 --no-cpu-baseline --no-1m --no-moving --no-extra > gpurun_out/m1_$name.json 2>/dev/null )
  python - <<EOF2
import json
d=json.loads(open("gpurun_out/m1_$name.json").read().strip().splitlines()[-1])
print("$name", "1M frame %.4f ms" % d["ms_per_step"], {k:round(v["ms_per_step"],3) for k,v in d["kernels"].items() if v["ms_per_step"]>0.004})
EOF2
done
