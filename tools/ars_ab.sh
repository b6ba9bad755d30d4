#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
timeout 900 python -m pytest tests -m gpu -x -q -k "active_ray or fused_ba or next_rows" 2>&1 | tail -3
for v in 1 0; do
  NARUTO_DEBUG_ARS_FUSED=$v timeout 300 python bench.py --workload office0_ba_iter --active-ray 2>/dev/null | grep '^{"metric' | python -c "
import json,sys; d=json.loads(sys.stdin.readline()); print('ARS_FUSED=$v', d['ms_per_step'], d.get('pieces_eager_ms'))"
done
cd /tmp && export TMPDIR=/tmp; R=${GRAFT_REPO_ROOT}
timeout 300 rocprofv3 --kernel-trace -d $R/gpurun_out/q_ars -o kt -- python $R/bench.py --workload office0_ba_iter --active-ray > /dev/null 2>&1
python $R/tools/prof_summary.py $(find $R/gpurun_out/q_ars -name "*.db" | head -1) 2>/dev/null | head -12 | cut -c1-160
rm -rf $R/gpurun_out/q_ars
