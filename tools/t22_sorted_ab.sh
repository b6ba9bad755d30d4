#!/bin/bash
# the T = 2^22 workload (BASELINE configs[4], one GPU's shard): Morton-ordered forward (round 6, default) against the packed forward (NARUTO_FWD_SORTED=0),
# step time + per-kernel times from a rocprofv3 kernel trace
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R; mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
for v in 1 0; do
  rm -rf $R/gpurun_out/t22_kt
  NARUTO_FWD_SORTED=$v timeout 400 rocprofv3 --kernel-trace -d $R/gpurun_out/t22_kt -o kt -- python $R/bench.py --workload unit1024_T22_131072x43 --no-cpu-baseline --no-dropin --no-kernels --no-mapping-iter --steps ${STEPS:-60} --warmup ${WARM:-60} > $R/gpurun_out/t22_bench_$v.json 2> $R/gpurun_out/t22_kt.log
  python $R/tools/prof_summary.py $(find $R/gpurun_out/t22_kt -name "*.db" | head -1) > $R/gpurun_out/t22_trace_sorted$v.txt 2>/dev/null; rm -rf $R/gpurun_out/t22_kt
  echo "== NARUTO_FWD_SORTED=$v: $(grep -o '"ms_per_step[a-z_0-9]*": [0-9.]*' $R/gpurun_out/t22_bench_$v.json | head -2 | tr '\n' ' ')"
  head -16 $R/gpurun_out/t22_trace_sorted$v.txt | cut -c1-58,96-160
done
