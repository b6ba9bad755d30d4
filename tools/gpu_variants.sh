#!/bin/bash
# usage: gpu_variants.sh <lib1.so> <lib2.so> ...   -- bench (fp32 + bf16) each library variant under rocprofv3, print the top kernels
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R; mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
for lib in "$@"; do
  tag=$(basename $lib .so)
  for mode in fp32 bf16; do
    NARUTO_HIP_LIB=$R/$lib timeout 300 rocprofv3 --kernel-trace -d $R/gpurun_out/v_kt -o kt -- python $R/bench.py --mlp $mode --no-cpu-baseline --no-kernels --steps 50 > $R/gpurun_out/v_bench_${tag}_$mode.json 2> $R/gpurun_out/v_kt.log
    python $R/tools/prof_summary.py $(find $R/gpurun_out/v_kt -name "*.db" | head -1) > $R/gpurun_out/v_trace_${tag}_$mode.txt; rm -rf $R/gpurun_out/v_kt
    echo "== $tag $mode: $(cut -c1-20,130-175 $R/gpurun_out/v_bench_${tag}_$mode.json | tr -d '\n' | grep -o 'ms_per_step[^,]*')"
    head -5 $R/gpurun_out/v_trace_${tag}_$mode.txt | tail -4 | cut -c1-40,96-170
  done
done
