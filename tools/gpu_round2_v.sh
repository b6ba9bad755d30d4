#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R; mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/v_build.log 2>&1
cd /tmp && export TMPDIR=/tmp
for mode in fp32 bf16; do
  timeout 300 python $R/bench.py --mlp $mode --no-cpu-baseline --no-kernels --steps 50 | cut -c1-160
  NARUTO_DEBUG_NO_SIDE_BRANCH=1 timeout 300 python $R/bench.py --mlp $mode --no-cpu-baseline --no-kernels --steps 50 | cut -c1-160
done
timeout 300 rocprofv3 --kernel-trace -d $R/gpurun_out/v_kt -o kt -- python $R/bench.py --no-cpu-baseline --no-kernels --steps 50 > $R/gpurun_out/v_bench.json 2> $R/gpurun_out/v_kt.log
python $R/tools/prof_summary.py $(find $R/gpurun_out/v_kt -name "*.db" | head -1) > $R/gpurun_out/v_kernel_trace.txt; rm -rf $R/gpurun_out/v_kt
head -12 $R/gpurun_out/v_kernel_trace.txt | cut -c1-44,96-170
timeout 300 python $R/bench.py --no-graph --no-cpu-baseline --no-kernels --steps 50 | cut -c1-160
cd $R; timeout 1500 python -m pytest tests -m gpu -q -x -k "train or smooth or capture or trainer or dp or parallel or reproducible" 2>&1 | tail -3
