cd $GRAFT_REPO_ROOT
export NBLS_LIB=$PWD/narrow_band_least_squares_amd/csrc/libnbls_hip_dev.so
timeout -k 10 120 python tools/quick_time.py cfg3 1 2 screen_stamps=1 ablate=2048 screen_nopf=1 2>&1 | grep "stamps"
timeout -k 10 120 python tools/quick_time.py cfg3 1 2 screen_stamps=1 ablate=2048 2>&1 | grep "stamps"
timeout -k 10 120 python tools/quick_time.py cfg3 1 2 screen_stamps=1 ablate=2048 screen_b_dma=1 2>&1 | grep "stamps"
