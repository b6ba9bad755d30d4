cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for W in 8 4; do for wl in office0_2048x43 office0_2048x128; do for P in 1 2; do
  if [ $wl = office0_2048x43 ] && [ $P = 2 ]; then continue; fi
  if [ $wl = office0_2048x128 ] && [ $P = 1 ]; then P=0; fi
  echo "== W=$W wl=$wl PACKED=$P"
  NARUTO_PACK_WAVES=$W NARUTO_FWD_PACKED=$P rocprofv3 --kernel-trace --stats -d gpurun_out/pw_${W}_${wl}_$P -o x -- python bench.py --workload $wl --steps 200 --warmup 20 --no-cpu-baseline --no-mapping-iter > gpurun_out/pw_${W}_${wl}_$P.log 2>&1
  grep -h "k_query_fwd_loss" gpurun_out/pw_${W}_${wl}_$P/*kernel_stats.csv gpurun_out/pw_${W}_${wl}_$P/*/*kernel_stats.csv 2>/dev/null | cut -d, -f1-5 | cut -c1-160
  grep -h '^{"metric' gpurun_out/pw_${W}_${wl}_$P.log | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'])"
  rm -rf gpurun_out/pw_${W}_${wl}_$P
done; done; done
