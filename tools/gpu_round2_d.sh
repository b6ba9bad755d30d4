#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/d_build.log 2>&1
timeout 1800 python -m pytest tests -m gpu -q --maxfail=10 -k "bf16 or full_size_against or two_rank or capture or reference_loop or config0 or retained or large_tables" > gpurun_out/d_pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/d_pytest.log
tail -12 gpurun_out/d_pytest.log | cut -c1-300
for v in "" abl_NOSTAGE abl_NOATOMIC abl_NOTILE abl_NOEPI bwd1; do
  if [ -n "$v" ]; then export NARUTO_HIP_LIB=$R/naruto_amd/variants/libnaruto_hip_$v.so; else unset NARUTO_HIP_LIB; fi
  timeout 300 python bench.py --no-cpu-baseline --mlp bf16 --steps 20 > gpurun_out/d_bench_bf16_$v.json 2> gpurun_out/d_bench_bf16_$v.err
  python -c "
import json; d=json.loads(open('gpurun_out/d_bench_bf16_$v.json').read().strip().splitlines()[-1]); print('variant [$v]', d['ms_per_step'], [ (k['kernel'][:14], k['ms']) for k in d['kernels'] if k['kernel'].startswith('k_query_bwd') or k['kernel'].startswith('k_hash')])"
done
unset NARUTO_HIP_LIB
BENCH_ARGS="--mlp bf16" bash tools/pmc_sq.sh d_bf16 > gpurun_out/d_sq.log 2>&1
grep "k_query_bwd" gpurun_out/d_bf16_sq.txt | cut -c1-30,96-200
