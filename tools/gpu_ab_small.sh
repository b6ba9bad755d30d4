#!/bin/bash
# usage: gpu_ab_small.sh <lib.so|default>...  -- ms per step of the small / headline workloads for each library variant, 2 runs each
R=${GRAFT_REPO_ROOT:-$(pwd)}
for lib in "$@"; do
  for wl in office0_2048x43 office0_2048x128 office0_8192x43; do
    for rep in 1 2; do
      if [ "$lib" = default ]; then L=""; else L="NARUTO_HIP_LIB=$R/$lib"; fi
      out=$(env $L timeout 300 python $R/bench.py --workload $wl --no-cpu-baseline --no-kernels --no-dropin --steps 100 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])")
      echo "$(basename $lib .so) $wl $out"
    done
  done
done
