# Evidence for profiles/: run through gpurun from the repo root:  gpurun -- 'bash tools/profile_round.sh r01'
# 1./2. FETCH_SIZE / WRITE_SIZE PMC passes -> traffic.json, 3. bench JSON line (reads that traffic.json),
# 4. rocprofv3 kernel stats of the same command.
TAG=${1:-r01}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/prof
rm -rf gpurun_out/prof/stats gpurun_out/prof/fetch gpurun_out/prof/write
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/prof/fetch -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/prof/fetch.log 2>&1 || exit 3
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/prof/write -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/prof/write.log 2>&1 || exit 4
python tools/make_traffic.py ${TAG} gpurun_out/prof --traffic-only || exit 5
python bench.py --steps 5 --warmup 1 > gpurun_out/prof/${TAG}_bench_cfg3.json 2> gpurun_out/prof/bench.err || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof/stats -- python bench.py --steps 5 --warmup 1 --no-cpu-baseline > gpurun_out/prof/stats.log 2>&1 || exit 2
python tools/make_traffic.py ${TAG} gpurun_out/prof
