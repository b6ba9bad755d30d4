#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
timeout 1500 python -m pytest tests -m gpu -x -q -k "packed_forward or train_step or edge_sizes or fused_ba or two_rank or golden" 2>&1 | tail -3
for v in 1 0; do
  for wl in office0_ba_iter office0_2048x43 office0_8192x43; do
    extra="--no-cpu-baseline --no-dropin --no-kernels --no-mapping-iter"; [ $wl = office0_ba_iter ] && extra=""
    NARUTO_FWD_PACKED=$v timeout 300 python bench.py --workload $wl --steps 20 $extra 2>/dev/null | grep '^{"metric' | python -c "
import json,sys; d=json.loads(sys.stdin.readline()); print('PACKED=$v $wl', d['ms_per_step'], d.get('ms_per_step_median_of_5_chunks'))"
  done
  NARUTO_FWD_PACKED=$v timeout 300 python bench.py --workload office0_ba_iter --active-ray 2>/dev/null | grep '^{"metric' | python -c "
import json,sys; d=json.loads(sys.stdin.readline()); print('PACKED=$v ba active', d['ms_per_step'])"
done
