# Evidence for profiles/: run through gpurun from the repo root:  gpurun --timeout 1100 -- 'bash tools/profile_round.sh r03'
#   1./2. FETCH_SIZE / WRITE_SIZE PMC passes of the bench command (cfg-3) -> traffic.json (per kernel, per pass)
#   3.    bench JSON line of every configuration (cfg-3 reads that traffic.json)
#   4.    rocprofv3 --kernel-trace --stats of the same bench command per configuration
TAG=${1:-r04}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/prof_$TAG
rm -rf $OUT && mkdir -p $OUT
BENCH="python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-noise"
timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch -- $BENCH > $OUT/fetch.log 2>&1 || exit 3
timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/write -- $BENCH > $OUT/write.log 2>&1 || exit 4
python tools/make_traffic.py $TAG $OUT --traffic-only || exit 5
python bench.py --entry ltsva --steps 10 --warmup 3 > $OUT/${TAG}_bench_ltsva_cfg3.json 2> $OUT/bench_ltsva.err || exit 6
python bench.py --config cfg1b --steps 20 --warmup 3 --no-noise > $OUT/${TAG}_bench_cfg1b.json 2> $OUT/bench_cfg1b.err || exit 7
for CFG in cfg3 cfg2 cfg5 cfg4; do
  python bench.py --config $CFG --steps 6 --warmup 2 > $OUT/${TAG}_bench_$CFG.json 2> $OUT/bench_$CFG.err || exit 1
  timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_$CFG -- python bench.py --config $CFG --steps 4 --warmup 1 --no-cpu-baseline --no-noise > $OUT/stats_$CFG.log 2>&1 || exit 2
  cp $OUT/stats_$CFG/*/*kernel_stats.csv $OUT/${TAG}_bench_${CFG}_kernel_stats.csv
  echo "$CFG done: $(tail -c 300 $OUT/${TAG}_bench_$CFG.json | head -c 10)"
done
bash tools/pmc_traffic.sh cfg3 1 > $OUT/${TAG}_pmc_traffic_per_pass.txt 2>&1
bash tools/pmc_screen.sh 1 > $OUT/${TAG}_sq_counters.txt 2>&1
python tools/call_breakdown.py 2>&1 | grep -v CAUTION > $OUT/${TAG}_call_breakdown.txt
# the large-array FAST-LTS kernel (solve_bucket.inc): SQ counters + instructions per wave, developer phase stamps
bash tools/pmc_lts.sh cfg5 1 > $OUT/${TAG}_lts_sq_counters_cfg5.txt 2>&1
bash tools/pmc_lts.sh cfg4 0.25 bands=12 > $OUT/${TAG}_lts_sq_counters_cfg4.txt 2>&1
if [ -f narrow_band_least_squares_amd/csrc/libnbls_hip_dev.so ]; then
  for C in "cfg5 1" "cfg4 0.25 bands=12"; do
    NBLS_LIB=narrow_band_least_squares_amd/csrc/libnbls_hip_dev.so python tools/quick_time.py $C 2 lts_stamps=1 2>&1 | grep -v CAUTION | tail -n 4
  done > $OUT/${TAG}_lts_stamps.txt
fi
bash tools/pmc_insts.sh 1 > $OUT/${TAG}_insts_per_wave.txt 2>&1
if [ -f narrow_band_least_squares_amd/csrc/libnbls_hip_dev.so ]; then
  NBLS_LIB=narrow_band_least_squares_amd/csrc/libnbls_hip_dev.so python tools/quick_time.py cfg3 1 3 lts_stamps=1 2>&1 | grep -v CAUTION | tail -n 4 > $OUT/${TAG}_lts_wave_stamps_cfg3.txt
fi
python tools/long_window_time.py 2>&1 | grep -v CAUTION > $OUT/${TAG}_long_window_time.txt
python tools/call_timeline.py cfg3 2>&1 | grep -v CAUTION > $OUT/${TAG}_call_timeline.txt
for n in 8 4 2; do python tools/sharded_rehearsal.py $n 2>&1 | tail -n 1; NBLS_STREAM_RESULTS=0 python tools/sharded_rehearsal.py $n 2>&1 | tail -n 1; done > $OUT/${TAG}_sharded_rehearsal.txt
[ -x tools/lds_atomic_rate ] && ./tools/lds_atomic_rate > $OUT/${TAG}_lds_atomic_rate.txt 2>&1
[ -x tools/bk_pass_rate ] && ./tools/bk_pass_rate > $OUT/${TAG}_bk_pass_rate.txt 2>&1
[ -x tools/valu_rate ] && ./tools/valu_rate > $OUT/${TAG}_valu_rate.txt 2>&1
ls $OUT
