#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/c_build.log 2>&1
timeout 2400 python -m pytest tests -m gpu -q --maxfail=10 > gpurun_out/c_pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/c_pytest.log
tail -25 gpurun_out/c_pytest.log | cut -c1-400
timeout 600 python tools/bf16_error_study.py > gpurun_out/c_bf16_study.txt 2>&1; cat gpurun_out/c_bf16_study.txt | grep -v Warn
for mode in fp32 bf16; do
  NARUTO_BENCH_MLP=$mode timeout 600 python bench.py --no-cpu-baseline --mlp $mode > gpurun_out/c_bench_$mode.json 2> gpurun_out/c_bench_$mode.err
  echo "bench $mode rc=$?"; python -c "
import json; d=json.loads(open('gpurun_out/c_bench_$mode.json').read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value']); [print('   %-55s %9.4f'%(k['kernel'][:55],k['ms'])) for k in d['kernels']]"
done
NARUTO_HIP_LIB=$R/naruto_amd/variants/libnaruto_hip_bwd1.so timeout 600 python bench.py --no-cpu-baseline --mlp bf16 > gpurun_out/c_bench_bf16_bwd1.json 2> gpurun_out/c_bench_bf16_bwd1.err
python -c "
import json; d=json.loads(open('gpurun_out/c_bench_bf16_bwd1.json').read().strip().splitlines()[-1]); print('bwd 1 wave/SIMD variant', d['ms_per_step'], d['value']); [print('   %-55s %9.4f'%(k['kernel'][:55],k['ms'])) for k in d['kernels']]"
timeout 900 python bench.py --workload unit1024_T22_131072x43 --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/c_bench_T22.json 2> gpurun_out/c_bench_T22.err
python -c "
import json; d=json.loads(open('gpurun_out/c_bench_T22.json').read().strip().splitlines()[-1]); print('T22', d['ms_per_step'], d['value']); [print('   %-55s %9.4f'%(k['kernel'][:55],k['ms'])) for k in d['kernels']]"
