#!/bin/bash
# the data-parallel step at world size 1 (real RCCL calls, one rank): step time per launch mode + the kernel trace of the default mode
cd ${GRAFT_REPO_ROOT:-.}
for mode in segmented whole eager; do
  NARUTO_FORCE_DIST=1 NARUTO_GRAPH_DIST=$mode timeout 300 python bench.py --steps 50 --no-cpu-baseline --no-dropin --no-kernels --no-mapping-iter 2>/dev/null | grep '^{"metric' | python -c "
import json,sys; d=json.loads(sys.stdin.readline()); print('$mode', d['ms_per_step'], d.get('ms_per_step_median_of_5_chunks'), d['config'].get('hip_graph'))"
done
cd /tmp && export TMPDIR=/tmp; R=${GRAFT_REPO_ROOT}
NARUTO_FORCE_DIST=1 timeout 300 rocprofv3 --kernel-trace -d $R/gpurun_out/q_dp -o kt -- python $R/bench.py --steps 50 --no-cpu-baseline --no-dropin --no-kernels --no-mapping-iter > /dev/null 2>&1
python $R/tools/prof_summary.py $(find $R/gpurun_out/q_dp -name "*.db" | head -1) 2>/dev/null | head -24 | cut -c1-170
rm -rf $R/gpurun_out/q_dp
