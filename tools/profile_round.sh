#!/bin/bash
# Collect the rocprofv3 evidence for one workload on the GPU box (run through gpurun from the repo root):
#   bash tools/profile_round.sh r02 office0_2048x128 [steps]
#     -> gpurun_out/<tag>_<workload>_{kernel_trace,pmc_FETCH_SIZE,pmc_WRITE_SIZE}.txt (+ the bench JSON line under rocprof)
# Kernel trace: the default bench command (hipGraph replay).  PMC: separate passes, eager launches (--no-graph) so that every
# dispatch is attributed to its kernel, kernel table included (the roofline kernel's own launch shape); never combined with trace
# domains other than --kernel-trace.
set -u
TAG=${1:-r02}
W=${2:-office0_2048x128}
STEPS=${3:-20}
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
mkdir -p $R/gpurun_out
P=$R/gpurun_out/${TAG}_${W}
timeout 900 rocprofv3 --kernel-trace -d ${P}_kt -o kt -- python $R/bench.py --workload $W --steps $STEPS --warmup 5 --no-cpu-baseline --no-dropin --no-mapping-iter > ${P}_bench_under_rocprof.json 2> ${P}_kt.log
python $R/tools/prof_summary.py $(find ${P}_kt -name "*.db" | head -1) > ${P}_kernel_trace.txt
rm -rf ${P}_kt
for c in FETCH_SIZE WRITE_SIZE TCC_REQ_sum; do
  timeout 900 rocprofv3 --kernel-trace --pmc $c -d ${P}_pmc_$c -o pmc -- python $R/bench.py --workload $W --no-graph --no-cpu-baseline --no-dropin --no-mapping-iter --steps 6 --warmup ${PMC_WARMUP:-2} > /dev/null 2> ${P}_pmc_$c.log
  python $R/tools/prof_summary.py $(find ${P}_pmc_$c -name "*.db" | head -1) > ${P}_pmc_$c.txt
  rm -rf ${P}_pmc_$c
done
head -30 ${P}_kernel_trace.txt | cut -c1-60,96-170
