#!/bin/bash
# usage: gpu_trace_variants.sh <workload> <lib.so|default>...  -- kernel trace (top kernels) of one workload per library variant
R=${GRAFT_REPO_ROOT:-$(pwd)}
W=$1; shift
cd /tmp && export TMPDIR=/tmp
for lib in "$@"; do
  tag=$(basename $lib .so)
  if [ "$lib" = default ]; then L=""; else L="NARUTO_HIP_LIB=$R/$lib"; fi
  env $L timeout 300 rocprofv3 --kernel-trace -d $R/gpurun_out/vt_$tag -o kt -- python $R/bench.py --workload $W --no-cpu-baseline --no-kernels --no-dropin --steps 50 > $R/gpurun_out/vt_$tag.json 2> $R/gpurun_out/vt_$tag.log
  echo "== $tag $W: $(python -c "import json;print(json.load(open('$R/gpurun_out/vt_$tag.json'))['ms_per_step'])")"
  python $R/tools/prof_summary.py $(find $R/gpurun_out/vt_$tag -name "*.db" | head -1) | head -8 | tail -7 | cut -c1-40,96-150
  rm -rf $R/gpurun_out/vt_$tag
done
