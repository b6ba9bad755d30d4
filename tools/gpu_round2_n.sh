#!/bin/bash
# per-level split counts of the dense scatter units: sweep under rocprofv3
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R; mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/n_build.log 2>&1
cd /tmp && export TMPDIR=/tmp
run() {
  tag=$1; shift
  env "$@" timeout 300 rocprofv3 --kernel-trace -d $R/gpurun_out/n_kt -o kt -- python $R/bench.py --no-cpu-baseline --no-kernels --steps 50 > $R/gpurun_out/n_bench_$tag.json 2> $R/gpurun_out/n_kt_$tag.log
  python $R/tools/prof_summary.py $(find $R/gpurun_out/n_kt -name "*.db" | head -1) > $R/gpurun_out/n_kernel_trace_$tag.txt; rm -rf $R/gpurun_out/n_kt
  echo "$tag: $(grep -h k_hash_scatter_lds $R/gpurun_out/n_kernel_trace_$tag.txt | head -1 | cut -c1-30,96-170)"
}
run base X=1
run l012_3_l34_5 NARUTO_DEBUG_SCATTER_SPLITS_LEVELS=0:3,1:3,2:3,3:5,4:5
run l0_2_l1_3_l2_4 NARUTO_DEBUG_SCATTER_SPLITS_LEVELS=0:2,1:3,2:4,3:5,4:5
run l34_5 NARUTO_DEBUG_SCATTER_SPLITS_LEVELS=3:5,4:5
run l4_5 NARUTO_DEBUG_SCATTER_SPLITS_LEVELS=4:5
run l012_5_l34_3 NARUTO_DEBUG_SCATTER_SPLITS_LEVELS=0:5,1:5,2:5,3:3,4:3
run l012_5 NARUTO_DEBUG_SCATTER_SPLITS_LEVELS=0:5,1:5,2:5
run l0_6_l1_5 NARUTO_DEBUG_SCATTER_SPLITS_LEVELS=0:6,1:5,2:5,3:4,4:3
timeout 900 python -m pytest $R/tests -m gpu -q -x -k "scatter or backward or train or reproducible" 2>&1 | tail -3
