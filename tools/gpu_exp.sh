#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R; mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build()" > /dev/null 2>&1
for w in office0_2048x128 office0_2048x43 office0_8192x43 mp3d_2048x256 unit1024_131072x43 unit1024_T22_131072x43; do timeout 600 python bench.py --mlp bf16 --workload $w --steps 10 --warmup 3 --no-cpu-baseline --no-kernels | cut -c85-160; done
timeout 900 python -m pytest tests -m gpu -q -x -k "bf16" 2>&1 | tail -3
