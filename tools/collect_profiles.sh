#!/bin/bash
# copy the judged evidence of tools/measure_round.sh from gpurun_out/ (scratch) into profiles/ (tracked)
set -eu
TAG=${1:-r02}
cd "$(dirname "$0")/.."
G=gpurun_out; P=profiles
for f in bench_default bench_dropin bench_ba_iter bench_ba_iter_active_ray bench_bf16 bench_eval_fp32 bench_eval_bf16 bench_office0_2048x43 bench_office0_8192x43 bench_mp3d_2048x256 bench_unit1024_131072x43 bench_T22_fp32 bench_T22_bf16; do
  [ -s $G/${TAG}_$f.json ] && tail -1 $G/${TAG}_$f.json | python -m json.tool > $P/${TAG}_$f.json
done
cp $G/${TAG}_bf16_error_study.txt $P/ 2>/dev/null || true
grep -v "amdgpu.ids" $G/${TAG}_bf16_error_study.txt > $P/${TAG}_bf16_error_study.txt 2>/dev/null || true
for w in office0_2048x128 unit1024_T22_131072x43; do
  for k in kernel_trace pmc_FETCH_SIZE pmc_WRITE_SIZE pmc_TCC_REQ_sum; do cp $G/${TAG}_${w}_$k.txt $P/ 2>/dev/null || true; done
  tail -1 $G/${TAG}_${w}_bench_under_rocprof.json | python -m json.tool > $P/${TAG}_${w}_bench_under_rocprof.json 2>/dev/null || true
done
cp $G/${TAG}_office0_2048x128_bf16_kernel_trace.txt $P/ 2>/dev/null || true
tcc() { [ -s $G/${TAG}_$1_pmc_TCC_REQ_sum.txt ] && echo ",$G/${TAG}_$1_pmc_TCC_REQ_sum.txt" || true; }
python tools/pmc_json.py office0_2048x128 $G/${TAG}_office0_2048x128_pmc_FETCH_SIZE.txt $G/${TAG}_office0_2048x128_pmc_WRITE_SIZE.txt$(tcc office0_2048x128) \
       unit1024_T22_131072x43 $G/${TAG}_unit1024_T22_131072x43_pmc_FETCH_SIZE.txt $G/${TAG}_unit1024_T22_131072x43_pmc_WRITE_SIZE.txt$(tcc unit1024_T22_131072x43) > $P/${TAG}_pmc.json
for f in dropin_kernel_trace dropin_torch_profiler_swap_only dropin_torch_profiler_fused_adam_fused_smoothness hbm_random_line_bench gather_valu_overlap_bench office0_2048x43_kernel_trace office0_ba_iter_kernel_trace; do
  [ -s $G/${TAG}_$f.txt ] && grep -v "amdgpu.ids\|UserWarning\|_warn_once\|ROCTracer" $G/${TAG}_$f.txt | cut -c1-220 > $P/${TAG}_$f.txt
done
for f in sq_counters_sq sq_counters_2048x43_sq; do [ -s $G/${TAG}_$f.txt ] && cp $G/${TAG}_$f.txt $P/${TAG}_${f%_sq}.txt; done
for f in fwd_timeline_2048x43 fwd_timeline_ba fwd_timeline_2048x128_packed_everywhere trained_step trained_step_flat accuracy_study short_timeline scatter_timeline fwd_lab walk_timeline x3_chain_stats t22_band_stats_sorted1 t22_band_stats_sorted0; do
  [ -s $G/${TAG}_$f.txt ] && grep -v "amdgpu.ids\|UserWarning\|_warn_once\|ROCTracer" $G/${TAG}_$f.txt | cut -c1-260 > $P/${TAG}_$f.txt
done
[ -s $G/${TAG}_accuracy_study.json ] && cp $G/${TAG}_accuracy_study.json $P/
git rev-parse HEAD > $P/${TAG}_commit.txt   # the tree the measurement ran on: commit before measuring, collect before the next commit
tail -3 $G/${TAG}_pytest_gpu.log > $P/${TAG}_pytest_gpu_summary.txt
ls -la $P | tail -30
