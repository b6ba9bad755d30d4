#!/bin/bash
# A/B of the training forward's launch shapes at sample counts the depth-ordered walk cannot take (32 + 11): the flat launch + k_loss_stage
# (NARUTO_FWD_PACKED=0), the packed forward (=3), the packed forward with every sample in one pass, and with two four-wave workgroups per CU --
# on the BA iteration (2 148 rays from the keyframe store), the benchmark batches and the HBM-resident table.   gpurun -- bash tools/ba_ab.sh
cd ${GRAFT_REPO_ROOT:-.}
for v in "0 0 8" "3 0 8" "3 1 8" "3 0 4"; do set -- $v
  for wl in office0_ba_iter office0_2048x43 office0_8192x43 unit1024_131072x43 unit1024_T22_131072x43; do
    extra="--no-cpu-baseline --no-dropin --no-kernels --no-mapping-iter"; [ $wl = office0_ba_iter ] && extra=""
    NARUTO_FWD_PACKED=$1 NARUTO_PACK_ONE_PASS=$2 NARUTO_PACK_WAVES=$3 timeout 300 python bench.py --workload $wl --steps 20 $extra 2>/dev/null | grep '^{"metric' | python -c "
import json,sys; d=json.loads(sys.stdin.readline()); print('PACKED=$1 ONE_PASS=$2 WAVES=$3 $wl', d['ms_per_step'], d.get('ms_per_step_median_of_5_chunks'))"
  done
  NARUTO_FWD_PACKED=$1 NARUTO_PACK_ONE_PASS=$2 NARUTO_PACK_WAVES=$3 timeout 300 python bench.py --workload office0_ba_iter --active-ray 2>/dev/null | grep '^{"metric' | python -c "
import json,sys; d=json.loads(sys.stdin.readline()); print('PACKED=$1 ONE_PASS=$2 WAVES=$3 office0_ba_iter --active-ray', d['ms_per_step'])"
done
