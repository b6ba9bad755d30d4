#!/bin/bash
# usage: gpu_variants_roles.sh <lib.so>...  -- k_hash_scatter_lds per role (1 dense, 2 hashed, 4 uncert, 7 all) for each library variant
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R; mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
for lib in "$@"; do
  tag=$(basename $lib .so)
  for role in 1 2 4 7; do
    NARUTO_DEBUG_SCATTER_ROLES=$role NARUTO_HIP_LIB=$R/$lib timeout 300 rocprofv3 --kernel-trace -d $R/gpurun_out/vr_kt -o kt -- python $R/bench.py --no-cpu-baseline --no-kernels --steps 30 > /dev/null 2> $R/gpurun_out/vr_kt.log
    python $R/tools/prof_summary.py $(find $R/gpurun_out/vr_kt -name "*.db" | head -1) > $R/gpurun_out/vr_trace_${tag}_$role.txt; rm -rf $R/gpurun_out/vr_kt
    echo "$tag role $role: $(grep k_hash_scatter_lds $R/gpurun_out/vr_trace_${tag}_$role.txt | head -1 | cut -c96-150)"
  done
done
