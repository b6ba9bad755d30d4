#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R; mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
for role in 1 2 3; do
  NARUTO_DEBUG_SAMPLE_ROLES=$role timeout 300 rocprofv3 --kernel-trace -d $R/gpurun_out/t_kt -o kt -- python $R/bench.py --no-cpu-baseline --no-kernels --steps 30 > /dev/null 2> $R/gpurun_out/t_kt.log
  python $R/tools/prof_summary.py $(find $R/gpurun_out/t_kt -name "*.db" | head -1) > $R/gpurun_out/t_trace_$role.txt; rm -rf $R/gpurun_out/t_kt
  echo "role $role: $(grep k_sample_encode $R/gpurun_out/t_trace_$role.txt | head -1 | cut -c1-30,96-170)"
done
