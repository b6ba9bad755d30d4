#!/bin/bash
# A/B of the training forward at sample counts that are not a multiple of 64 (round 5): NARUTO_WALK_PARTIAL=0 = flat tiles + k_sample_encode +
# k_loss_stage (the round-4 form), =2 = the depth-ordered walk with a partly filled last tile (five-launch iteration).  gpurun -- bash tools/walk_ab.sh
cd ${GRAFT_REPO_ROOT:-.}
for v in 0 2; do
  for wl in ${WLS:-office0_ba_iter office0_2048x43 office0_8192x43 unit1024_131072x43}; do
    extra="--no-cpu-baseline --no-dropin --no-kernels --no-mapping-iter"; [ $wl = office0_ba_iter ] && extra=""
    NARUTO_WALK_PARTIAL=$v timeout 300 python bench.py --workload $wl --steps 20 $extra 2>/dev/null | grep '^{"metric' | python -c "
import json,sys; d=json.loads(sys.stdin.readline()); print('WALK_PARTIAL=$v $wl', d['ms_per_step'], d.get('ms_per_step_median_of_5_chunks'))"
  done
  NARUTO_WALK_PARTIAL=$v timeout 300 python bench.py --workload office0_ba_iter --active-ray 2>/dev/null | grep '^{"metric' | python -c "
import json,sys; d=json.loads(sys.stdin.readline()); print('WALK_PARTIAL=$v office0_ba_iter --active-ray', d['ms_per_step'])"
done
