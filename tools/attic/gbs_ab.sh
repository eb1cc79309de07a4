#!/bin/bash
# per variant: 1M trace → group_bounds durations; 10M bench class times
cd $GRAFT_REPO_ROOT
for v in base gbs2 gbs8; do
  lib=broadphase_rs_amd/lib_ab/$v/libbroadphase_amd.so; [ "$v" = base ] && lib=broadphase_rs_amd/lib/libbroadphase_amd.so
  for n in 1000000 10000000; do
  (cd /tmp && export TMPDIR=/tmp && BP_LIBRARY=$GRAFT_REPO_ROOT/$lib rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/tr_$v -o t -- python3 $GRAFT_REPO_ROOT/tools/trace_frame.py $n 8 > /dev/null 2>&1)
  python tools/trace_show.py gpurun_out/tr_$v | grep "group_bounds\|frame" | sed "s/^/$v $n: /"
  rm -rf gpurun_out/tr_$v
  done
done
