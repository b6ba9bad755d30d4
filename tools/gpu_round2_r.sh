#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R; mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/r_build.log 2>&1
timeout 2400 python -m pytest tests -m gpu -q --maxfail=5 2>&1 | tail -8
cd /tmp && export TMPDIR=/tmp
export MASTER_ADDR=127.0.0.1 MASTER_PORT=29511 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0
for gd in eager segmented; do
  NARUTO_GRAPH_DIST=$gd NARUTO_FORCE_DIST=1 timeout 300 python $R/bench.py --no-cpu-baseline --no-kernels --steps 50 | cut -c1-160
  NARUTO_GRAPH_DIST=$gd NARUTO_FORCE_DIST=1 timeout 300 rocprofv3 --kernel-trace -d $R/gpurun_out/r_kt -o kt -- python $R/bench.py --no-cpu-baseline --no-kernels --steps 50 > $R/gpurun_out/r_bench_dp_$gd.json 2> $R/gpurun_out/r_kt.log
  python $R/tools/prof_summary.py $(find $R/gpurun_out/r_kt -name "*.db" | head -1) > $R/gpurun_out/r_trace_dp_$gd.txt; rm -rf $R/gpurun_out/r_kt
  head -16 $R/gpurun_out/r_trace_dp_$gd.txt | cut -c1-44,96-170
done
