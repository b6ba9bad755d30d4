#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
timeout 2400 python -m pytest tests -m gpu -x -q -k "launch_variants or train_step or mapping_iterations or trainer or graph or capture or golden or reference_workload or full_size or tail_in_the_backward or dropin or two_rank or accuracy" 2>&1 | tail -4
for v in 1 0; do
  NARUTO_TV_MOVE=$v timeout 300 python bench.py --steps 100 --no-cpu-baseline --no-dropin --no-kernels --no-mapping-iter 2>/dev/null | grep '^{"metric' | python -c "
import json,sys; d=json.loads(sys.stdin.readline()); print('TV_MOVE=$v fp32', d['ms_per_step'], d.get('ms_per_step_median_of_5_chunks'))"
  NARUTO_TV_MOVE=$v timeout 300 python bench.py --mlp bf16 --steps 100 --no-cpu-baseline --no-dropin --no-kernels --no-mapping-iter 2>/dev/null | grep '^{"metric' | python -c "
import json,sys; d=json.loads(sys.stdin.readline()); print('TV_MOVE=$v bf16', d['ms_per_step'], d.get('ms_per_step_median_of_5_chunks'))"
done
