#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/b_build.log 2>&1
timeout 2400 python -m pytest tests -m gpu -q --maxfail=10 > gpurun_out/b_pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/b_pytest.log
tail -25 gpurun_out/b_pytest.log | cut -c1-300
bash tools/profile_round.sh r02 unit1024_T22_131072x43 8 > gpurun_out/b_prof_T22.log 2>&1
tail -32 gpurun_out/b_prof_T22.log
