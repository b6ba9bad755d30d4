#!/bin/bash
# SQ counters of k_hash_scatter_lds per role (NARUTO_DEBUG_SCATTER_ROLES: 1 dense units, 2 hashed units, 4 uncertainty-grid units)
#   bash tools/pmc_scatter_roles.sh <tag>  ->  gpurun_out/<tag>_roles.txt
set -u
TAG=${1:-roles}
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
: > $R/gpurun_out/${TAG}_roles.txt
for role in 1 2 4; do
i=0
while read -r line; do
  [ -z "$line" ] && continue
  i=$((i+1))
  NARUTO_DEBUG_SCATTER_ROLES=$role timeout 150 rocprofv3 --kernel-trace --pmc $line -d $R/gpurun_out/${TAG}_p -o pmc -- python $R/bench.py --no-graph --no-cpu-baseline --no-kernels --steps 10 --warmup 3 > /dev/null 2> $R/gpurun_out/${TAG}_p.log
  echo "role $role: $line" >> $R/gpurun_out/${TAG}_roles.txt
  python $R/tools/prof_summary.py $(find $R/gpurun_out/${TAG}_p -name "*.db" | head -1) | grep "k_hash_scatter" | grep -v "calls" | cut -c1-30,96-220 >> $R/gpurun_out/${TAG}_roles.txt
  rm -rf $R/gpurun_out/${TAG}_p
done <<'LIST'
SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM
SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS
SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU
SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE
SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_INSTS_BRANCH SQ_ACTIVE_INST_SCA
LIST
done
cat $R/gpurun_out/${TAG}_roles.txt
