cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for ab in 0 32 1 2 4 5 7; do rm -rf gpurun_out/pa; NBLS_ABLATE=$ab timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/pa -- python tools/quick_time.py cfg3 0.25 3 1 > /dev/null 2>&1; echo "ablate=$ab $(grep screen_kernel gpurun_out/pa/*/*kernel_stats.csv | cut -d, -f2-4)"; done
