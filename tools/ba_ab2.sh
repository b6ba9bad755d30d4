#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
for v in 1 3; do
  for wl in office0_ba_iter office0_2048x43 office0_8192x43 unit1024_T22_131072x43; do
    extra="--no-cpu-baseline --no-dropin --no-kernels --no-mapping-iter"; [ $wl = office0_ba_iter ] && extra=""
    NARUTO_FWD_PACKED=$v timeout 300 python bench.py --workload $wl --steps 20 $extra 2>/dev/null | grep '^{"metric' | python -c "
import json,sys; d=json.loads(sys.stdin.readline()); print('PACKED=$v $wl', d['ms_per_step'], d.get('ms_per_step_median_of_5_chunks'))"
  done
done
