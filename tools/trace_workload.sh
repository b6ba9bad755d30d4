#!/bin/bash
# kernel trace (rocprofv3 --kernel-trace) of one bench workload under hipGraph replay -> gpurun_out/<tag>_<workload>_kernel_trace.txt
#   bash tools/trace_workload.sh r03 office0_2048x43 [extra bench args]
set -u
TAG=$1; W=$2; shift 2
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
mkdir -p $R/gpurun_out
P=$R/gpurun_out/${TAG}_${W}
timeout 600 rocprofv3 --kernel-trace -d ${P}_kt -o kt -- python $R/bench.py --workload $W --steps 50 --warmup 5 --no-cpu-baseline --no-kernels --no-dropin "$@" > ${P}_bench_under_rocprof.json 2> ${P}_kt.log
python $R/tools/prof_summary.py $(find ${P}_kt -name "*.db" | head -1) > ${P}_kernel_trace.txt
rm -rf ${P}_kt
head -14 ${P}_kernel_trace.txt | cut -c1-50,96-170
