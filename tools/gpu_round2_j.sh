#!/bin/bash
# full parity suite + the round's measurement set (bench lines + rocprofv3 summaries)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build(); g.smoke()" > gpurun_out/j_build_smoke.log 2>&1; tail -2 gpurun_out/j_build_smoke.log
timeout 2400 python -m pytest tests -m gpu -q --maxfail=10 > gpurun_out/j_pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/j_pytest.log
tail -6 gpurun_out/j_pytest.log | cut -c1-300
timeout 900 python bench.py > gpurun_out/j_bench_default.json 2> gpurun_out/j_bench_default.err; cut -c1-300 gpurun_out/j_bench_default.json
timeout 600 python bench.py --mlp bf16 --no-cpu-baseline > gpurun_out/j_bench_bf16.json 2> gpurun_out/j_bench_bf16.err; cut -c1-200 gpurun_out/j_bench_bf16.json
timeout 600 python bench.py --workload office0_8192x43_eval --no-cpu-baseline > gpurun_out/j_bench_eval.json 2> gpurun_out/j_bench_eval.err; cut -c1-1500 gpurun_out/j_bench_eval.json
timeout 600 python bench.py --workload office0_8192x43_eval --mlp bf16 --no-cpu-baseline > gpurun_out/j_bench_eval_bf16.json 2> gpurun_out/j_bench_eval_bf16.err; cut -c1-200 gpurun_out/j_bench_eval_bf16.json
for w in office0_2048x43 office0_8192x43 mp3d_2048x256 unit1024_131072x43; do
  timeout 600 python bench.py --workload $w --no-cpu-baseline --no-kernels --steps 20 > gpurun_out/j_bench_$w.json 2> gpurun_out/j_bench_$w.err; cut -c1-160 gpurun_out/j_bench_$w.json; echo
done
bash tools/profile_round.sh r02 office0_2048x128 30 > gpurun_out/j_prof_default.log 2>&1; tail -14 gpurun_out/j_prof_default.log
bash tools/profile_round.sh r02 unit1024_T22_131072x43 8 > gpurun_out/j_prof_T22.log 2>&1; tail -16 gpurun_out/j_prof_T22.log
