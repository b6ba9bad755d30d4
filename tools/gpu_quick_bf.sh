#!/bin/bash
# bf16-mode parity tests, then the headline bench in both MLP modes under rocprofv3 (top kernels)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R; mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q -k "${1:-bf16 or launch_variants or render_fused}" 2>&1 | tail -4
cd /tmp && export TMPDIR=/tmp
for mode in bf16 fp32; do
  rm -rf $R/gpurun_out/q_kt
  timeout 300 rocprofv3 --kernel-trace -d $R/gpurun_out/q_kt -o kt -- python $R/bench.py --mlp $mode --no-cpu-baseline --no-dropin --no-mapping-iter --steps 50 > $R/gpurun_out/q_bench_$mode.json 2> $R/gpurun_out/q_kt.log
  python $R/tools/prof_summary.py $(find $R/gpurun_out/q_kt -name "*.db" | head -1) > $R/gpurun_out/q_trace_$mode.txt; rm -rf $R/gpurun_out/q_kt
  echo "== $mode: $(grep -o '"ms_per_step": [0-9.]*' $R/gpurun_out/q_bench_$mode.json | head -1)"
  head -8 $R/gpurun_out/q_trace_$mode.txt | cut -c1-60,96-160
done
