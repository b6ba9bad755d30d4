#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/g_build.log 2>&1
timeout 1200 python -m pytest tests -m gpu -q --maxfail=10 -k "render_fused or golden" > gpurun_out/g_pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/g_pytest.log
tail -5 gpurun_out/g_pytest.log | cut -c1-300
run() { # label roles dense hashed uncert
  export NARUTO_DEBUG_SCATTER_ROLES=$2
  if [ -n "$3" ]; then export NARUTO_DEBUG_SCATTER_SPLITS_DENSE=$3; else unset NARUTO_DEBUG_SCATTER_SPLITS_DENSE; fi
  if [ -n "$4" ]; then export NARUTO_DEBUG_SCATTER_SPLITS_HASHED=$4; else unset NARUTO_DEBUG_SCATTER_SPLITS_HASHED; fi
  if [ -n "$5" ]; then export NARUTO_DEBUG_SCATTER_SPLITS_UNCERT=$5; else unset NARUTO_DEBUG_SCATTER_SPLITS_UNCERT; fi
  timeout 300 python bench.py --no-cpu-baseline --steps 30 > gpurun_out/g_$1.json 2> gpurun_out/g_$1.err
  python -c "
import json,sys; d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print('%-26s step %.4f ms  '%(sys.argv[2], d['ms_per_step']), [(k['kernel'][:10], k['ms']) for k in d['kernels'] if k['kernel'].startswith('k_hash')])" gpurun_out/g_$1.json "$1"
}
run all_d4_h2_u2 7 4 2 2
run none 0 4 2 2
run dense_only_d5 1 5 2 2
run dense_only_d4 1 4 2 2
run dense_only_d3 1 3 2 2
run hashed_only_h2 2 4 2 2
run hashed_only_h3 2 4 3 2
run uncert_only_u1 4 4 2 1
run uncert_only_u2 4 4 2 2
run uncert_only_u4 4 4 2 4
run uncert_only_u8 4 4 2 8
run all_d3_h2_u4 7 3 2 4
run all_d4_h2_u4 7 4 2 4
run all_d3_h2_u8 7 3 2 8
