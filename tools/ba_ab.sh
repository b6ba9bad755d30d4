#!/bin/bash
# A/B of the training forward's launch shapes on the BA iteration (2 148 rays x 43) and the large 43-sample batches
cd ${GRAFT_REPO_ROOT:-.}
for v in "0 8" "1 8" "1 4"; do set -- $v
  for wl in office0_ba_iter office0_2048x43 unit1024_131072x43; do
    extra="--no-cpu-baseline --no-dropin --no-kernels --no-mapping-iter"; [ $wl = office0_ba_iter ] && extra=""
    NARUTO_FWD_PACKED=$1 NARUTO_PACK_WAVES=$2 timeout 300 python bench.py --workload $wl --steps 20 $extra 2>/dev/null | grep '^{"metric' | python -c "
import json,sys; d=json.loads(sys.stdin.readline()); print('PACKED=$1 W=$2 $wl', d['ms_per_step'], d.get('mapping_iter_ms'))"
  done
done
NARUTO_FWD_PACKED=1 timeout 300 python bench.py --workload office0_ba_iter --active-ray 2>/dev/null | grep '^{"metric' | cut -c1-200
timeout 300 python bench.py --workload mp3d_2048x256 --steps 20 --no-cpu-baseline --no-dropin --no-mapping-iter 2>/dev/null | grep '^{"metric' | python -c "
import json,sys; d=json.loads(sys.stdin.readline()); print('mp3d', d['ms_per_step'])"
