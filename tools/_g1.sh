set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "variants or filter or cfg1 or segment" 2>&1 | tail -2
for i in 1 2; do
timeout -k 10 120 python tools/quick_time.py cfg3 1 3 2>&1 | grep "^cfg3" | tail -1 | sed 's/^/T16 /'
NBLS_LIB=$PWD/narrow_band_least_squares_amd/csrc/libnbls_hip_T32.so timeout -k 10 120 python tools/quick_time.py cfg3 1 3 2>&1 | grep "^cfg3" | tail -1 | sed 's/^/T32 /'
done
timeout -k 10 200 python tools/quick_time.py cfg4 0.1 2 2>&1 | grep "^cfg4" | tail -1 | sed 's/^/T16 /'
NBLS_LIB=$PWD/narrow_band_least_squares_amd/csrc/libnbls_hip_T32.so timeout -k 10 200 python tools/quick_time.py cfg4 0.1 2 2>&1 | grep "^cfg4" | tail -1 | sed 's/^/T32 /'
