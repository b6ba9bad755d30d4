#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R; mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/w_build.log 2>&1
timeout 600 python bench.py --no-cpu-baseline > gpurun_out/w_bench.json 2> gpurun_out/w_bench.err; tail -3 gpurun_out/w_bench.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/w_bench.json').read().strip().splitlines()[-1])
print(d['ms_per_step'], json.dumps(d['roofline']))
for r in d['kernels']:
    if 'iteration' in r['kernel']: print(r)
PY
timeout 600 python bench.py --no-cpu-baseline --workload unit1024_T22_131072x43 --steps 5 --warmup 2 > gpurun_out/w_bench_T22.json 2> gpurun_out/w_bench_T22.err; tail -2 gpurun_out/w_bench_T22.err; cut -c1-300 gpurun_out/w_bench_T22.json
python -m pytest tests -m gpu -q -x -k "exports or abi or symbol" 2>&1 | tail -2
