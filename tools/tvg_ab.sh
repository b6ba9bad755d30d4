#!/bin/bash
# level groups per lattice-encode workgroup in the training forward's tail role (NARUTO_TV_TAIL_GROUPS = 1 | 2 | 4): headline, shipped sampling, BA iteration
cd ${GRAFT_REPO_ROOT:-.}
for g in 1 2 4; do
  for wl in office0_2048x128 office0_2048x43 office0_ba_iter; do
    extra="--no-cpu-baseline --no-dropin --no-kernels --no-mapping-iter"; [ $wl = office0_ba_iter ] && extra=""
    NARUTO_TV_TAIL_GROUPS=$g timeout 300 python bench.py --workload $wl --steps 20 $extra 2>/dev/null | grep '^{"metric' | python -c "
import json,sys; d=json.loads(sys.stdin.readline()); print('TV_TAIL_GROUPS=$g $wl', d['ms_per_step'], d.get('ms_per_step_median_of_5_chunks'))"
  done
done
