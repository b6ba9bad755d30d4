#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
for st in 0 1 2 3 4 257 258 259 260; do
  NARUTO_DEBUG_WALK_STAGGER=$st timeout 300 python bench.py --steps 100 --no-cpu-baseline --no-dropin --no-mapping-iter 2>/dev/null | grep '^{"metric' | python -c "
import json,sys; d=json.loads(sys.stdin.readline()); print('stagger=$st', d['ms_per_step'], d.get('ms_per_step_median_of_5_chunks'), d['roofline_gather']['kernel_ms_in_iteration'])"
done
