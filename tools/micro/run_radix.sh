#!/bin/bash
# tools/micro/run_radix.sh <variant|base>[:ENV=..,ENV=..] ...   -> one radix_bench run per spec (lib_ab/<variant> or lib)
cd "$(dirname "$0")/../.."
for spec in "$@"; do
  v=${spec%%:*}; envs=""; [ "$spec" != "$v" ] && envs=${spec#*:}
  dir=$PWD/broadphase_rs_amd/lib_ab/$v; [ "$v" = base ] && dir=$PWD/broadphase_rs_amd/lib
  (
    IFS=','; for kv in $envs; do [ -n "$kv" ] && export "$kv"; done
    echo "== $spec"; LD_LIBRARY_PATH=$dir:/opt/rocm/lib timeout -k 10 120 tools/micro/radix_bench ${RB_ARGS}
  ) 2>&1
done
