#!/bin/bash
# The round's measurement set on the GPU box (through gpurun, from the repo root):
#   bash tools/measure_round.sh r02      -> gpurun_out/<tag>_*  (copy what is to be judged into profiles/ with tools/collect_profiles.sh)
# parity suite + smoke, the bench lines (default / bf16 / eval / other workloads / T=2^22), rocprofv3 kernel traces and the
# FETCH_SIZE / WRITE_SIZE passes for the default and the T=2^22 workload, the bf16 error study.
set -u
TAG=${1:-r02}
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build(); g.smoke()" > gpurun_out/${TAG}_build_smoke.log 2>&1; tail -1 gpurun_out/${TAG}_build_smoke.log
timeout 2400 python -m pytest tests -m gpu -q --maxfail=10 > gpurun_out/${TAG}_pytest_gpu.log 2>&1
echo "pytest rc=$?" >> gpurun_out/${TAG}_pytest_gpu.log
tail -4 gpurun_out/${TAG}_pytest_gpu.log | cut -c1-300
timeout 900 python bench.py > gpurun_out/${TAG}_bench_default.json 2> gpurun_out/${TAG}_bench_default.err; cut -c1-200 gpurun_out/${TAG}_bench_default.json
timeout 600 python bench.py --mlp bf16 --no-cpu-baseline > gpurun_out/${TAG}_bench_bf16.json 2> /dev/null; cut -c1-200 gpurun_out/${TAG}_bench_bf16.json
for m in fp32 bf16; do timeout 600 python bench.py --workload office0_8192x43_eval --mlp $m --no-cpu-baseline --no-mapping-iter > gpurun_out/${TAG}_bench_eval_$m.json 2> /dev/null; cut -c1-200 gpurun_out/${TAG}_bench_eval_$m.json; done
for w in office0_2048x43 office0_8192x43 mp3d_2048x256 unit1024_131072x43; do
  timeout 600 python bench.py --workload $w --no-cpu-baseline --no-mapping-iter --steps 20 > gpurun_out/${TAG}_bench_$w.json 2> /dev/null; cut -c1-160 gpurun_out/${TAG}_bench_$w.json; echo
done
for m in fp32 bf16; do timeout 900 python bench.py --workload unit1024_T22_131072x43 --mlp $m --steps 60 --warmup 60 --no-cpu-baseline > gpurun_out/${TAG}_bench_T22_$m.json 2> /dev/null; cut -c1-160 gpurun_out/${TAG}_bench_T22_$m.json; echo; done
timeout 600 python tools/bf16_error_study.py > gpurun_out/${TAG}_bf16_error_study.txt 2>&1
bash tools/profile_round.sh $TAG office0_2048x128 30 > gpurun_out/${TAG}_prof_default.log 2>&1; tail -12 gpurun_out/${TAG}_prof_default.log
PMC_WARMUP=100 bash tools/profile_round.sh $TAG unit1024_T22_131072x43 8 > gpurun_out/${TAG}_prof_T22.log 2>&1; tail -10 gpurun_out/${TAG}_prof_T22.log
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace -d $R/gpurun_out/${TAG}_ktbf -o kt -- python $R/bench.py --mlp bf16 --no-cpu-baseline --steps 30 > /dev/null 2> $R/gpurun_out/${TAG}_ktbf.log
python $R/tools/prof_summary.py $(find $R/gpurun_out/${TAG}_ktbf -name "*.db" | head -1) > $R/gpurun_out/${TAG}_office0_2048x128_bf16_kernel_trace.txt; rm -rf $R/gpurun_out/${TAG}_ktbf
# round 3: the unchanged caller (bench --path dropin under rocprofv3 + torch.profiler tables), the mapping iteration end to end, the random-line ceiling
timeout 300 rocprofv3 --kernel-trace -d $R/gpurun_out/${TAG}_ktdrop -o kt -- python $R/bench.py --path dropin --steps 30 --warmup 10 > $R/gpurun_out/${TAG}_bench_dropin.json 2> $R/gpurun_out/${TAG}_ktdrop.log
python $R/tools/prof_summary.py $(find $R/gpurun_out/${TAG}_ktdrop -name "*.db" | head -1) > $R/gpurun_out/${TAG}_dropin_kernel_trace.txt; rm -rf $R/gpurun_out/${TAG}_ktdrop
cd $R
timeout 300 python tools/prof_dropin.py torch reference 8 > gpurun_out/${TAG}_dropin_torch_profiler_swap_only.txt 2>&1
timeout 300 python tools/prof_dropin.py fused fused 8 > gpurun_out/${TAG}_dropin_torch_profiler_fused_adam_fused_smoothness.txt 2>&1
timeout 300 python bench.py --workload office0_ba_iter > gpurun_out/${TAG}_bench_ba_iter.json 2> /dev/null; cut -c1-160 gpurun_out/${TAG}_bench_ba_iter.json; echo
timeout 300 python bench.py --workload office0_ba_iter --active-ray > gpurun_out/${TAG}_bench_ba_iter_active_ray.json 2> /dev/null; cut -c1-160 gpurun_out/${TAG}_bench_ba_iter_active_ray.json; echo
[ -x tools/hrl_bench ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/hbm_random_line_bench.hip -o tools/hrl_bench > /dev/null 2>&1
[ -x tools/gvo_bench ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/gather_valu_overlap_bench.hip -o tools/gvo_bench > /dev/null 2>&1
timeout 120 tools/hrl_bench > gpurun_out/${TAG}_hbm_random_line_bench.txt 2>&1
timeout 120 tools/gvo_bench > gpurun_out/${TAG}_gather_valu_overlap_bench.txt 2>&1
bash tools/trace_workload.sh $TAG office0_2048x43 > /dev/null 2>&1
bash tools/trace_workload.sh $TAG office0_ba_iter --active-ray > /dev/null 2>&1
bash tools/pmc_sq.sh ${TAG}_sq_counters > /dev/null 2>&1
BENCH_ARGS="--workload office0_2048x43" bash tools/pmc_sq.sh ${TAG}_sq_counters_2048x43 > /dev/null 2>&1
# round 4: per-step timeline of the packed training forward, the iteration on a trained map, the reconstruction-accuracy study (HIP fp32 / bf16 /
# NaN depths against the CPU oracle on the analytic room: ~2 min of CPU for the oracle's 390 iterations)
cd $R
NARUTO_FWD_PACKED=3 timeout 300 python tools/fwd_timeline.py office0_2048x43 > gpurun_out/${TAG}_fwd_timeline_2048x43.txt 2>&1
TIMELINE_GRAPH=1 NARUTO_FWD_PACKED=3 timeout 300 python tools/fwd_timeline_ba.py > gpurun_out/${TAG}_fwd_timeline_ba.txt 2>&1
NARUTO_FWD_PACKED=2 timeout 300 python tools/fwd_timeline.py office0_2048x128 > gpurun_out/${TAG}_fwd_timeline_2048x128_packed_everywhere.txt 2>&1
NARUTO_FWD_PACKED=3 timeout 600 python tools/time_trained_step.py > gpurun_out/${TAG}_trained_step.txt 2>&1
timeout 600 python tools/time_trained_step.py > gpurun_out/${TAG}_trained_step_flat.txt 2>&1
timeout 1500 python tests/accuracy_study.py --out gpurun_out/${TAG}_accuracy_study.json > gpurun_out/${TAG}_accuracy_study.txt 2>&1
# round 5: per-phase timelines of the short-ray forward and of the scatter's workgroups, the forward lab (fp32 chain against the x3 chain)
timeout 300 python tools/short_timeline.py 2148 > gpurun_out/${TAG}_short_timeline.txt 2>&1
timeout 300 python tools/scatter_timeline.py > gpurun_out/${TAG}_scatter_timeline.txt 2>&1
[ -x tools/fwd_lab ] && timeout 300 tools/fwd_lab > gpurun_out/${TAG}_fwd_lab.txt 2>&1
git -C $R rev-parse HEAD > gpurun_out/${TAG}_commit.txt 2>/dev/null || true
# round 6: per-wave timeline of the depth-ordered walk (trained state), the x3-vs-fp32 chain distance, what each large-table forward evaluates
cd $R
timeout 300 python tools/walk_timeline.py office0_2048x128 fp32 600 > gpurun_out/${TAG}_walk_timeline.txt 2>&1
timeout 300 python tools/x3_chain_stats.py > gpurun_out/${TAG}_x3_chain_stats.txt 2>&1
for v in 1 0; do NARUTO_FWD_SORTED=$v timeout 300 python tools/t22_band_stats.py 131072 150 > gpurun_out/${TAG}_t22_band_stats_sorted$v.txt 2>&1; done
