#!/bin/bash
# first GPU pass of round 2: parity suite, default bench line, the T=2^22 workloads
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/a_build.log 2>&1
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/a_pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/a_pytest.log
tail -15 gpurun_out/a_pytest.log
timeout 600 python bench.py > gpurun_out/a_bench_default.json 2> gpurun_out/a_bench_default.err
echo "bench rc=$?"; cut -c1-600 gpurun_out/a_bench_default.json
timeout 600 python bench.py --workload unit1024_T22_16384x43 --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/a_bench_T22_16k.json 2> gpurun_out/a_bench_T22_16k.err
echo "T22 16k rc=$?"; cut -c1-400 gpurun_out/a_bench_T22_16k.json; tail -3 gpurun_out/a_bench_T22_16k.err
timeout 900 python bench.py --workload unit1024_T22_131072x43 --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/a_bench_T22.json 2> gpurun_out/a_bench_T22.err
echo "T22 rc=$?"; cut -c1-400 gpurun_out/a_bench_T22.json; tail -3 gpurun_out/a_bench_T22.err
