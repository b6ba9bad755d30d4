#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
timeout 1800 python -m pytest tests -m gpu -x -q -k "active_ray or fused_ba or next_rows or packed_forward or train_step or edge_sizes or launch_variants or large_tables or two_rank" 2>&1 | tail -3
for wl in office0_ba_iter office0_2048x43; do
  extra="--no-cpu-baseline --no-dropin --no-kernels --no-mapping-iter"; [ $wl = office0_ba_iter ] && extra=""
  timeout 300 python bench.py --workload $wl --steps 20 $extra 2>/dev/null | grep '^{"metric' | python -c "
import json,sys; d=json.loads(sys.stdin.readline()); print('$wl', d['ms_per_step'], d.get('ms_per_step_median_of_5_chunks'))"
done
timeout 300 python bench.py --workload office0_ba_iter --active-ray 2>/dev/null | grep '^{"metric' | python -c "
import json,sys; d=json.loads(sys.stdin.readline()); print('ba active', d['ms_per_step'])"
