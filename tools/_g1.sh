set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "variants or any_array_size or hostile or cfg1 or product_against" > gpurun_out/t1.log 2>&1 || { tail -30 gpurun_out/t1.log; exit 1; }
tail -3 gpurun_out/t1.log
timeout -k 10 120 python tools/quick_time.py cfg3 1 3 2>&1 | grep "^cfg3" | tail -2
timeout -k 10 120 python tools/quick_time.py cfg3 1 3 noise=1 2>&1 | grep "^cfg3" | tail -1
timeout -k 10 120 python tools/quick_time.py cfg4 0.1 2 2>&1 | grep "^cfg4" | tail -1
export NBLS_LIB=$PWD/narrow_band_least_squares_amd/csrc/libnbls_hip_dev.so
timeout -k 10 120 python tools/quick_time.py cfg3 1 2 screen_stamps=1 ablate=2048 2>&1 | grep "stamps"
