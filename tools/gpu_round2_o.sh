#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R; mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/o_build.log 2>&1
timeout 900 python -m pytest tests -m gpu -q -x -k "tail_in_the_backward or ray_buffers or keyframe_store or device_rng or capture_mid" 2>&1 | tail -15
cd /tmp && export TMPDIR=/tmp
for mode in fp32 bf16; do
  timeout 300 python $R/bench.py --mlp $mode --no-cpu-baseline --no-kernels --steps 50 | cut -c1-160
  timeout 300 rocprofv3 --kernel-trace -d $R/gpurun_out/o_kt -o kt -- python $R/bench.py --mlp $mode --no-cpu-baseline --no-kernels --steps 50 > $R/gpurun_out/o_bench_$mode.json 2> $R/gpurun_out/o_kt_$mode.log
  python $R/tools/prof_summary.py $(find $R/gpurun_out/o_kt -name "*.db" | head -1) > $R/gpurun_out/o_kernel_trace_$mode.txt; rm -rf $R/gpurun_out/o_kt
  head -12 $R/gpurun_out/o_kernel_trace_$mode.txt | cut -c1-44,96-170
done
NARUTO_DEBUG_NO_FUSED_TAIL=1 timeout 300 python $R/bench.py --no-cpu-baseline --no-kernels --steps 50 | cut -c1-160
