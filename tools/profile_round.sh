#!/bin/bash
# Collect the rocprofv3 evidence for one round on the GPU box (run through gpurun from the repo root):
#   bash tools/profile_round.sh r01   ->  gpurun_out/<tag>_{kernel_trace,pmc_fetch,pmc_write}.txt (+ the bench JSON lines)
# Kernel trace: the default bench command (hipGraph replay).  PMC: separate passes, eager launches (--no-graph) so
# that every dispatch is attributed to its kernel, kernel table included (the roofline kernel's own launch shape); never combined with trace domains other than --kernel-trace.
set -u
TAG=${1:-r01}
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
mkdir -p $R/gpurun_out
rocprofv3 --kernel-trace -d $R/gpurun_out/${TAG}_kt -o kt -- python $R/bench.py --no-cpu-baseline > $R/gpurun_out/${TAG}_bench_under_rocprof.json 2> $R/gpurun_out/${TAG}_kt.log
python $R/tools/prof_summary.py $(find $R/gpurun_out/${TAG}_kt -name "*.db" | head -1) > $R/gpurun_out/${TAG}_kernel_trace.txt
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c -d $R/gpurun_out/${TAG}_pmc_$c -o pmc -- python $R/bench.py --no-graph --no-cpu-baseline --steps 10 --warmup 3 > /dev/null 2> $R/gpurun_out/${TAG}_pmc_$c.log
  python $R/tools/prof_summary.py $(find $R/gpurun_out/${TAG}_pmc_$c -name "*.db" | head -1) > $R/gpurun_out/${TAG}_pmc_$c.txt
done
tail -1 $R/gpurun_out/${TAG}_bench_under_rocprof.json | cut -c1-300
