#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R; mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/m_build.log 2>&1
cd /tmp && export TMPDIR=/tmp
for mode in fp32 bf16; do
  timeout 300 python $R/bench.py --mlp $mode --no-cpu-baseline --no-kernels --steps 50 | cut -c1-160
  timeout 300 rocprofv3 --kernel-trace -d $R/gpurun_out/m_kt -o kt -- python $R/bench.py --mlp $mode --no-cpu-baseline --no-kernels --steps 50 > $R/gpurun_out/m_bench_$mode.json 2> $R/gpurun_out/m_kt_$mode.log
  python $R/tools/prof_summary.py $(find $R/gpurun_out/m_kt -name "*.db" | head -1) > $R/gpurun_out/m_kernel_trace_$mode.txt; rm -rf $R/gpurun_out/m_kt
  head -5 $R/gpurun_out/m_kernel_trace_$mode.txt | cut -c1-44,96-170
done
for w in office0_8192x43 mp3d_2048x256 unit1024_131072x43 unit1024_T22_131072x43; do timeout 600 python $R/bench.py --workload $w --steps 10 --warmup 3 --no-cpu-baseline --no-kernels | cut -c1-160; done
timeout 300 rocprofv3 --kernel-trace -d $R/gpurun_out/m_kt -o kt -- python $R/bench.py --workload unit1024_T22_131072x43 --steps 8 --warmup 3 --no-cpu-baseline --no-kernels > /dev/null 2> $R/gpurun_out/m_kt_T22.log
python $R/tools/prof_summary.py $(find $R/gpurun_out/m_kt -name "*.db" | head -1) > $R/gpurun_out/m_kernel_trace_T22.txt; rm -rf $R/gpurun_out/m_kt
head -8 $R/gpurun_out/m_kernel_trace_T22.txt | cut -c1-44,96-170
