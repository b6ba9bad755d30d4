#!/bin/bash
# usage: gpu_quick.sh [pytest -k expression]   -- the forward's parity tests, then a short bench of the headline + 2048x43 under rocprofv3 (top kernels)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R; mkdir -p gpurun_out
K=${1:-"train_step or golden or query_fwd or launch_variants or render"}
timeout 1500 python -m pytest tests -m gpu -x -q -k "$K" 2>&1 | tail -8 > gpurun_out/q_pytest.txt
cat gpurun_out/q_pytest.txt
cd /tmp && export TMPDIR=/tmp
for wl in office0_2048x128 office0_2048x43; do
  rm -rf $R/gpurun_out/q_kt
  timeout 300 rocprofv3 --kernel-trace -d $R/gpurun_out/q_kt -o kt -- python $R/bench.py --workload $wl --no-cpu-baseline --no-dropin --no-mapping-iter --steps 50 > $R/gpurun_out/q_bench_$wl.json 2> $R/gpurun_out/q_kt.log
  python $R/tools/prof_summary.py $(find $R/gpurun_out/q_kt -name "*.db" | head -1) > $R/gpurun_out/q_trace_$wl.txt; rm -rf $R/gpurun_out/q_kt
  echo "== $wl: $(grep -o '"ms_per_step": [0-9.]*' $R/gpurun_out/q_bench_$wl.json | head -1)"
  head -9 $R/gpurun_out/q_trace_$wl.txt | cut -c1-60,96-160
done
