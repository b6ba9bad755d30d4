#!/bin/bash
# k_hash_scatter_lds and the step under different scatter plans: one "VAR=val VAR=val" environment per line on stdin
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R; mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
while read -r envs; do
  ms=$(env $envs timeout 300 rocprofv3 --kernel-trace -d $R/gpurun_out/sp_kt -o kt -- python $R/bench.py ${BENCH_ARGS:-} --no-cpu-baseline --no-kernels --steps 30 2> $R/gpurun_out/sp_kt.log | tail -1 | python -c "import json,sys; print(json.loads(sys.stdin.read())['ms_per_step'])")
  python $R/tools/prof_summary.py $(find $R/gpurun_out/sp_kt -name "*.db" | head -1) > $R/gpurun_out/sp_trace.txt; rm -rf $R/gpurun_out/sp_kt
  echo "[$envs] step $ms ms; scatter: $(grep k_hash_scatter_lds $R/gpurun_out/sp_trace.txt | head -1 | cut -c96-150)"
done
