#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
for v in "3 1" "3 0" "0 0"; do set -- $v
  for wl in office0_ba_iter office0_2048x43 office0_8192x43; do
    extra="--no-cpu-baseline --no-dropin --no-kernels --no-mapping-iter"; [ $wl = office0_ba_iter ] && extra=""
    NARUTO_FWD_PACKED=$1 NARUTO_PACK_ONE_PASS=$2 timeout 300 python bench.py --workload $wl --steps 20 $extra 2>/dev/null | grep '^{"metric' | python -c "
import json,sys; d=json.loads(sys.stdin.readline()); print('PACKED=$1 ONE_PASS=$2 $wl', d['ms_per_step'], d.get('ms_per_step_median_of_5_chunks'))"
  done
done
NARUTO_FWD_PACKED=3 NARUTO_PACK_ONE_PASS=1 timeout 600 python -m pytest tests -m gpu -x -q -k "packed_forward or train_step_direct or edge_sizes or random_shapes" 2>&1 | tail -2
