#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/f_build.log 2>&1
timeout 1200 python -m pytest tests -m gpu -q --maxfail=10 -k "render_fused or bitwise or capture or golden or direct_against or two_rank or query_backward" > gpurun_out/f_pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/f_pytest.log
tail -8 gpurun_out/f_pytest.log | cut -c1-300
run() { # label lib dense_splits
  if [ -n "$2" ]; then export NARUTO_HIP_LIB=$R/naruto_amd/variants/libnaruto_hip_$2.so; else unset NARUTO_HIP_LIB; fi
  if [ -n "$3" ]; then export NARUTO_DEBUG_SCATTER_SPLITS_DENSE=$3; else unset NARUTO_DEBUG_SCATTER_SPLITS_DENSE; fi
  timeout 300 python bench.py --no-cpu-baseline --steps 30 > gpurun_out/f_$1.json 2> gpurun_out/f_$1.err
  python -c "
import json,sys; d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print('%-22s step %.4f ms  '%(sys.argv[2], d['ms_per_step']), [(k['kernel'][:10], k['ms']) for k in d['kernels'] if k['kernel'].startswith('k_hash') or k['kernel'].startswith('k_query_bwd')])" gpurun_out/f_$1.json "$1"
}
run run8_auto "" ""
run run8_d5 "" 5
run run8_d3 "" 3
run run16_auto run16 ""
run run16_d5 run16 5
run run16_d3 run16 3
run run32_auto run32 ""
run run32_d3 run32 3
run run32_d2 run32 2
