#!/bin/bash
cd /tmp && export TMPDIR=/tmp; R=${GRAFT_REPO_ROOT}
NARUTO_FWD_PACKED=3 timeout 300 rocprofv3 --kernel-trace -d $R/gpurun_out/q_bas -o kt -- python $R/bench.py --workload office0_ba_iter > /dev/null 2>&1
DB=$(find $R/gpurun_out/q_bas -name "*.db" | head -1)
python $R/tools/kernel_series.py $DB k_query_fwd_loss_packed
python $R/tools/kernel_series.py $DB k_hash_scatter_lds
rm -rf $R/gpurun_out/q_bas
NARUTO_FWD_PACKED=3 timeout 300 rocprofv3 --kernel-trace -d $R/gpurun_out/q_bas -o kt -- python $R/bench.py --workload office0_2048x43 --no-cpu-baseline --no-dropin --no-kernels --no-mapping-iter > /dev/null 2>&1
DB=$(find $R/gpurun_out/q_bas -name "*.db" | head -1)
python $R/tools/kernel_series.py $DB k_query_fwd_loss_packed
rm -rf $R/gpurun_out/q_bas
