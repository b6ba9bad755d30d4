#!/bin/bash
# usage: gpu_ab_workloads.sh <lib.so>...  -- ms per step of four workloads for each library variant (NARUTO_HIP_LIB)
R=${GRAFT_REPO_ROOT:-$(pwd)}
for lib in "$@"; do
  for wl in office0_2048x128 office0_8192x43 mp3d_2048x256 unit1024_131072x43; do
    out=$(NARUTO_HIP_LIB=$R/$lib timeout 300 python $R/bench.py --workload $wl --no-cpu-baseline --no-kernels --steps 50 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])")
    echo "$(basename $lib .so) $wl $out"
  done
done
