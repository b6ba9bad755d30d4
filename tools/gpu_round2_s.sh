#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R; mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/s_build.log 2>&1
timeout 900 python -m pytest tests -m gpu -q -x -k "sample_z or render or golden or device_rng or train_step" 2>&1 | tail -5
cd /tmp && export TMPDIR=/tmp
for mode in fp32; do
  timeout 300 python $R/bench.py --mlp $mode --no-cpu-baseline --no-kernels --steps 50 | cut -c1-160
  timeout 300 rocprofv3 --kernel-trace -d $R/gpurun_out/s_kt -o kt -- python $R/bench.py --mlp $mode --no-cpu-baseline --no-kernels --steps 50 > $R/gpurun_out/s_bench_$mode.json 2> $R/gpurun_out/s_kt_$mode.log
  python $R/tools/prof_summary.py $(find $R/gpurun_out/s_kt -name "*.db" | head -1) > $R/gpurun_out/s_kernel_trace_$mode.txt; rm -rf $R/gpurun_out/s_kt
  head -10 $R/gpurun_out/s_kernel_trace_$mode.txt | cut -c1-44,96-170
done
timeout 300 python $R/bench.py --workload office0_8192x43_eval --no-cpu-baseline --no-kernels | cut -c1-200
