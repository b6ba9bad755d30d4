#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R; mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
for v in 1 2 4; do
  NARUTO_HIP_LIB=$R/tools/scratch/libnaruto_tv$v.so timeout 300 rocprofv3 --kernel-trace -d $R/gpurun_out/u_kt -o kt -- python $R/bench.py --no-cpu-baseline --no-kernels --steps 30 > $R/gpurun_out/u_bench_$v.json 2> $R/gpurun_out/u_kt.log
  python $R/tools/prof_summary.py $(find $R/gpurun_out/u_kt -name "*.db" | head -1) > $R/gpurun_out/u_trace_$v.txt; rm -rf $R/gpurun_out/u_kt
  echo "tv levels $v: $(grep k_sample_encode $R/gpurun_out/u_trace_$v.txt | head -1 | cut -c1-30,96-170) | fwd $(grep k_query_fwd $R/gpurun_out/u_trace_$v.txt | head -1 | cut -c110-125) | $(grep -o 'ms_per_step[^,]*' $R/gpurun_out/u_bench_$v.json)"
done
cd $R; timeout 900 python -m pytest tests -m gpu -q -x -k "smooth or tv or train_step or golden" 2>&1 | tail -3
