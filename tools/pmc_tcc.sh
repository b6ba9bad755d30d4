#!/bin/bash
# L2 (TCC) counter passes over the eager bench: how many requests reach the L2 per launch of the forward, and how busy it is.
#   [BENCH_ARGS="--mlp bf16"] bash tools/pmc_tcc.sh <tag>   ->  gpurun_out/<tag>_tcc.txt
set -u
TAG=${1:-tcc}
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
: > $R/gpurun_out/${TAG}_tcc.txt
i=0
while read -r line; do
  [ -z "$line" ] && continue
  i=$((i+1))
  timeout 150 rocprofv3 --kernel-trace --pmc $line -d $R/gpurun_out/${TAG}_tcc_$i -o pmc -- python $R/bench.py --no-graph --no-cpu-baseline --no-kernels --steps 10 --warmup 3 ${BENCH_ARGS:-} > /dev/null 2> $R/gpurun_out/${TAG}_tcc_$i.log
  python $R/tools/prof_summary.py $(find $R/gpurun_out/${TAG}_tcc_$i -name "*.db" | head -1) | grep "k_query_fwd\|k_query_bwd\|k_hash_scatter\|k_sample_encode\|k_bwd_finish" | grep -v "calls" >> $R/gpurun_out/${TAG}_tcc.txt
  rm -rf $R/gpurun_out/${TAG}_tcc_$i
done <<'LIST'
TCC_REQ_sum TCC_READ_sum
TCC_HIT_sum TCC_MISS_sum
TCC_BUSY_sum GRBM_GUI_ACTIVE
TCC_TAG_STALL_sum TCC_WRITE_sum
TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum
LIST
cut -c1-34,96-200 $R/gpurun_out/${TAG}_tcc.txt
