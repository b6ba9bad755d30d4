#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R; mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build()" > /dev/null 2>&1
cd /tmp && export TMPDIR=/tmp
for mode in fp32 bf16; do
  timeout 300 python $R/bench.py --mlp $mode --no-cpu-baseline --no-kernels --steps 50 | grep -o 'ms_per_step[^,]*'
  NARUTO_DEBUG_NO_FUSED_LOSS_STAGE=1 timeout 300 python $R/bench.py --mlp $mode --no-cpu-baseline --no-kernels --steps 50 | grep -o 'ms_per_step[^,]*'
  timeout 300 rocprofv3 --kernel-trace -d $R/gpurun_out/e_kt -o kt -- python $R/bench.py --mlp $mode --no-cpu-baseline --no-kernels --steps 30 > $R/gpurun_out/e_bench.json 2> $R/gpurun_out/e_kt.log
  python $R/tools/prof_summary.py $(find $R/gpurun_out/e_kt -name "*.db" | head -1) > $R/gpurun_out/e_trace.txt; rm -rf $R/gpurun_out/e_kt
  head -8 $R/gpurun_out/e_trace.txt | cut -c1-40,96-150
done
timeout 300 python $R/bench.py --workload mp3d_2048x256 --no-cpu-baseline --no-kernels --steps 20 | grep -o 'ms_per_step[^,]*'
cd $R; timeout 2400 python -m pytest tests -m gpu -q --maxfail=5 2>&1 | tail -4
