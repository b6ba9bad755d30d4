#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
for v in 0 1; do
  NARUTO_EXP_TV_MOVE=$v timeout 300 python bench.py --steps 100 --no-cpu-baseline --no-dropin --no-kernels --no-mapping-iter 2>/dev/null | grep '^{"metric' | python -c "
import json,sys; d=json.loads(sys.stdin.readline()); print('EXP_TV_MOVE=$v', d['ms_per_step'], d.get('ms_per_step_median_of_5_chunks'))"
done
cd /tmp && export TMPDIR=/tmp; R=${GRAFT_REPO_ROOT}
NARUTO_EXP_TV_MOVE=1 timeout 300 rocprofv3 --kernel-trace -d $R/gpurun_out/q_tv -o kt -- python $R/bench.py --steps 100 --no-cpu-baseline --no-dropin --no-kernels --no-mapping-iter > /dev/null 2>&1
python $R/tools/prof_summary.py $(find $R/gpurun_out/q_tv -name "*.db" | head -1) 2>/dev/null | head -9 | cut -c1-170
rm -rf $R/gpurun_out/q_tv
