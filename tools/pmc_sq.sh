#!/bin/bash
# SQ counter passes (TA_*/TCP_* '_sum' counters crashed rocprofv3 on this pool: left out) over the eager bench (one pass per line of counters; --kernel-trace only, as the pool requires).
#   [BENCH_ARGS="--mlp bf16"] bash tools/pmc_sq.sh <tag>   ->  gpurun_out/<tag>_sq.txt
set -u
TAG=${1:-sq}
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
: > $R/gpurun_out/${TAG}_sq.txt
i=0
while read -r line; do
  [ -z "$line" ] && continue
  i=$((i+1))
  timeout 150 rocprofv3 --kernel-trace --pmc $line -d $R/gpurun_out/${TAG}_sq_$i -o pmc -- python $R/bench.py --no-graph --no-cpu-baseline --no-kernels --no-dropin --no-mapping-iter --steps 10 --warmup 3 ${BENCH_ARGS:-} > /dev/null 2> $R/gpurun_out/${TAG}_sq_$i.log
  python $R/tools/prof_summary.py $(find $R/gpurun_out/${TAG}_sq_$i -name "*.db" | head -1) | grep "k_query_fwd\|k_query_bwd\|k_hash_scatter" | grep -v "calls" >> $R/gpurun_out/${TAG}_sq.txt
  rm -rf $R/gpurun_out/${TAG}_sq_$i
done <<'LIST'
SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM
SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC
SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS
SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_VMEM_RD SQ_INSTS_SALU
SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_IFETCH SQ_IFETCH_LEVEL
SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQC_ICACHE_MISSES SQC_ICACHE_REQ
LIST
cat $R/gpurun_out/${TAG}_sq.txt | cut -c1-30,96-200
