#!/bin/bash
# does k_hash_scatter_lds's static SGPR-spill count (127 without the uncertainty units' scan + compaction, 307 with it) cost any
# workgroup type anything?  Per-role timelines of both builds (tools/libnaruto_hip_unc_compact0.so = -DNARUTO_UNC_COMPACT=0) at three batch sizes.
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
for wl in office0_2048x128 office0_8192x43 unit1024_131072x43; do
  for lib in default tools/libnaruto_hip_unc_compact0.so; do
    echo "=== $wl, library: $lib"
    if [ "$lib" = default ]; then timeout 300 python tools/scatter_timeline.py $wl 2>/dev/null | python tools/scatter_roles.py; else NARUTO_HIP_LIB=$R/$lib timeout 300 python tools/scatter_timeline.py $wl 2>/dev/null | python tools/scatter_roles.py; fi
  done
done
