#!/bin/bash
# usage: xcd_ab.sh [workload]   -- the XCD-partitioned gather in front of the training forward (round 6): off / group counts / boundaries, step time + the
# forward launches' kernel times from a rocprofv3 kernel trace
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R; mkdir -p gpurun_out
WL=${1:-office0_2048x128}
cd /tmp && export TMPDIR=/tmp
for v in "NARUTO_FWD_XCD_SPLIT=0" "NARUTO_XCD_G=2" "NARUTO_XCD_G=4" "NARUTO_XCD_G=4 NARUTO_XCD_GROUPS=7,10,13" "NARUTO_XCD_G=4 NARUTO_XCD_GROUPS=8,11,14" "NARUTO_XCD_G=8" "NARUTO_XCD_G=1"; do
  rm -rf $R/gpurun_out/x_kt
  env $v timeout 300 rocprofv3 --kernel-trace -d $R/gpurun_out/x_kt -o kt -- python $R/bench.py --workload $WL --no-cpu-baseline --no-dropin --no-mapping-iter --steps 100 > $R/gpurun_out/x_bench.json 2> $R/gpurun_out/x_kt.log
  python $R/tools/prof_summary.py $(find $R/gpurun_out/x_kt -name "*.db" | head -1) > $R/gpurun_out/x_trace.txt; rm -rf $R/gpurun_out/x_kt
  echo "== $WL [$v]: $(grep -o '"ms_per_step": [0-9.]*' $R/gpurun_out/x_bench.json | head -1)"
  grep -E "k_gather|k_query_fwd_loss|k_sample" $R/gpurun_out/x_trace.txt | cut -c1-40,64-130
done
