#!/bin/bash
# launch-to-launch timeline of the last dispatches of a workload (rocprofv3 kernel trace + tools/iter_timeline.py): usage iter_gaps.sh <out> <n> <bench args...>
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$1; N=$2; shift 2
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/it_kt
rocprofv3 --kernel-trace -d $R/gpurun_out/it_kt -o kt -- python $R/bench.py "$@" > /dev/null 2>&1
python $R/tools/iter_timeline.py $(find $R/gpurun_out/it_kt -name "*.db" | head -1) $N > $R/gpurun_out/$OUT 2>&1
rm -rf $R/gpurun_out/it_kt
