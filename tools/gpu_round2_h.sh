#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
mkdir -p $R/gpurun_out
run() { # label roles dense hashed uncert
  export NARUTO_DEBUG_SCATTER_ROLES=$2
  if [ -n "$3" ]; then export NARUTO_DEBUG_SCATTER_SPLITS_DENSE=$3; else unset NARUTO_DEBUG_SCATTER_SPLITS_DENSE; fi
  if [ -n "$4" ]; then export NARUTO_DEBUG_SCATTER_SPLITS_HASHED=$4; else unset NARUTO_DEBUG_SCATTER_SPLITS_HASHED; fi
  if [ -n "$5" ]; then export NARUTO_DEBUG_SCATTER_SPLITS_UNCERT=$5; else unset NARUTO_DEBUG_SCATTER_SPLITS_UNCERT; fi
  timeout 200 rocprofv3 --kernel-trace -d $R/gpurun_out/h_kt -o kt -- python $R/bench.py --no-cpu-baseline --no-kernels --steps 30 > $R/gpurun_out/h_$1.json 2> $R/gpurun_out/h_$1.log
  python $R/tools/prof_summary.py $(find $R/gpurun_out/h_kt -name "*.db" | head -1) > $R/gpurun_out/h_$1.txt; rm -rf $R/gpurun_out/h_kt
  echo "$1: step $(python -c "import json;print(json.loads(open('$R/gpurun_out/h_$1.json').read().strip().splitlines()[-1])['ms_per_step'])")  $(grep -E 'k_hash_scatter_lds|k_query_bwd|k_bwd_finish' $R/gpurun_out/h_$1.txt | head -3 | awk '{printf "%s=%s ", substr($1,12,14), $4}')"
}
run r1like_d5_h2_noU 3 5 2 2
run d4_h2_u2 7 4 2 2
run d3_h2_u2 7 3 2 2
run d3_h2_u4 7 3 2 4
run d4_h2_u3 7 4 2 3
run d3_h2_u3 7 3 2 3
run d4_h2_u1 7 4 2 1
run uonly_u2 4 4 2 2
run uonly_u4 4 4 2 4
run donly_d4 1 4 2 2
run donly_d3 1 3 2 2
run honly_h2 2 4 2 2
