#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R; mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
for v in br512 br1024; do
  NARUTO_HIP_LIB=$R/tools/scratch/libnaruto_$v.so timeout 600 rocprofv3 --kernel-trace -d $R/gpurun_out/x_kt -o kt -- python $R/bench.py --workload unit1024_T22_131072x43 --no-cpu-baseline --no-kernels --steps 8 --warmup 3 > $R/gpurun_out/x_bench_$v.json 2> $R/gpurun_out/x_kt.log
  python $R/tools/prof_summary.py $(find $R/gpurun_out/x_kt -name "*.db" | head -1) > $R/gpurun_out/x_trace_$v.txt; rm -rf $R/gpurun_out/x_kt
  echo "$v: $(grep -o 'ms_per_step[^,]*' $R/gpurun_out/x_bench_$v.json)"; grep "k_bin_fill\|k_bin_apply\|k_bin_count" $R/gpurun_out/x_trace_$v.txt | head -3 | cut -c1-30,96-160
done
cd $R; NARUTO_HIP_LIB=$R/tools/scratch/libnaruto_br1024.so timeout 900 python -m pytest tests -m gpu -q -x -k "large_tables or T22 or binned" 2>&1 | tail -3
