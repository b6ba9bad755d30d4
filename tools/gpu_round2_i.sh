#!/bin/bash
# SQ counters of k_hash_scatter_lds with only one unit type doing work
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
mkdir -p $R/gpurun_out
rocprofv3 -L 2>/dev/null | grep -o "SQ_LDS[A-Z_]*\|SQ_INSTS_LDS[A-Z_]*\|SQ_ACTIVE_INST_LDS\|SQ_WAIT_INST_LDS\|SQ_LDS_[A-Z_]*" | sort -u > $R/gpurun_out/i_lds_counters.txt
cat $R/gpurun_out/i_lds_counters.txt | tr '\n' ' '; echo
pass() { # label roles "counters"
  export NARUTO_DEBUG_SCATTER_ROLES=$2 NARUTO_DEBUG_SCATTER_SPLITS_DENSE=4 NARUTO_DEBUG_SCATTER_SPLITS_HASHED=2 NARUTO_DEBUG_SCATTER_SPLITS_UNCERT=2
  timeout 200 rocprofv3 --kernel-trace --pmc $3 -d $R/gpurun_out/i_p -o pmc -- python $R/bench.py --no-graph --no-cpu-baseline --no-kernels --steps 10 --warmup 3 > /dev/null 2> $R/gpurun_out/i_$1.log
  python $R/tools/prof_summary.py $(find $R/gpurun_out/i_p -name "*.db" | head -1) | grep "k_hash_scatter" | grep -v calls | awk -v L="$1" '{print L, $2, $3, $4}' >> $R/gpurun_out/i_sq.txt
  rm -rf $R/gpurun_out/i_p
}
: > $R/gpurun_out/i_sq.txt
for role in 1 2 4; do
  pass role$role $role "SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS"
  pass role$role $role "SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS"
  pass role$role $role "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU"
  pass role$role $role "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_ATOMIC_RETURN"
  pass role$role $role "SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC"
done
cat $R/gpurun_out/i_sq.txt
