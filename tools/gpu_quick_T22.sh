#!/bin/bash
# usage: gpu_quick_T22.sh   -- the large-table (binned scatter) parity tests, then the T = 2^22 workload under rocprofv3 (top kernels)
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R; mkdir -p gpurun_out
timeout 1200 python -m pytest tests -m gpu -x -q -k "large_tables or configs4 or T22 or scatter" 2>&1 | tail -4
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace -d $R/gpurun_out/q_kt22 -o kt -- python $R/bench.py --workload unit1024_T22_131072x43 --no-cpu-baseline --no-dropin --steps 8 --warmup 3 > $R/gpurun_out/q_bench_T22.json 2> $R/gpurun_out/q_kt22.log
python $R/tools/prof_summary.py $(find $R/gpurun_out/q_kt22 -name "*.db" | head -1) 2>/dev/null | head -16
rm -rf $R/gpurun_out/q_kt22
grep -h '^{"metric' $R/gpurun_out/q_bench_T22.json | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('ms_per_step', d['ms_per_step'])"
