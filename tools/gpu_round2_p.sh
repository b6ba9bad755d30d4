#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R; mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/p_build.log 2>&1
timeout 2400 python -m pytest tests -m gpu -q --maxfail=10 > gpurun_out/p_pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/p_pytest.log
tail -8 gpurun_out/p_pytest.log | cut -c1-300
cd /tmp && export TMPDIR=/tmp
for mode in fp32 bf16; do timeout 300 python $R/bench.py --mlp $mode --no-cpu-baseline --no-kernels --steps 50 | cut -c1-160; done
for w in office0_2048x43 office0_8192x43 mp3d_2048x256 unit1024_131072x43 unit1024_T22_131072x43; do timeout 600 python $R/bench.py --workload $w --steps 10 --warmup 3 --no-cpu-baseline --no-kernels | cut -c1-160; done
