#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/e_build.log 2>&1
timeout 2400 python -m pytest tests -m gpu -q --maxfail=10 > gpurun_out/e_pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/e_pytest.log
tail -12 gpurun_out/e_pytest.log | cut -c1-300
show() { python -c "
import json,sys; d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[2], d['ms_per_step'], d['value']); [print('   %-55s %9.4f'%(k['kernel'][:55],k['ms'])) for k in d['kernels']]" $1 "$2"; }
for mode in fp32 bf16; do
  timeout 600 python bench.py --no-cpu-baseline --mlp $mode > gpurun_out/e_bench_$mode.json 2> gpurun_out/e_bench_$mode.err; show gpurun_out/e_bench_$mode.json "bench $mode"
done
NARUTO_HIP_LIB=$R/naruto_amd/variants/libnaruto_hip_fwdbf2.so timeout 600 python bench.py --no-cpu-baseline --mlp bf16 > gpurun_out/e_bench_bf16_fwd2.json 2> gpurun_out/e_bench_bf16_fwd2.err; show gpurun_out/e_bench_bf16_fwd2.json "bf16 fwd at 2 waves/SIMD"
cd /tmp && export TMPDIR=/tmp
for mode in fp32 bf16; do
  timeout 300 rocprofv3 --kernel-trace -d $R/gpurun_out/e_kt_$mode -o kt -- python $R/bench.py --mlp $mode --no-cpu-baseline --no-kernels > /dev/null 2> $R/gpurun_out/e_kt_$mode.log
  python $R/tools/prof_summary.py $(find $R/gpurun_out/e_kt_$mode -name "*.db" | head -1) > $R/gpurun_out/e_kernel_trace_$mode.txt; rm -rf $R/gpurun_out/e_kt_$mode
  head -16 $R/gpurun_out/e_kernel_trace_$mode.txt | cut -c1-44,96-170
done
