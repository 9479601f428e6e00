set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 800 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -x -q -m gpu -k "variants or any_array_size or hostile or cfg4 or cfg5 or long" > gpurun_out/t1.log 2>&1 || { tail -30 gpurun_out/t1.log; exit 1; }
tail -3 gpurun_out/t1.log
for o in "" "screen_tb4=1"; do
timeout -k 10 200 python tools/quick_time.py cfg4 0.1 2 $o 2>&1 | grep "^cfg4" | tail -1
timeout -k 10 200 python tools/quick_time.py cfg5 1 2 $o 2>&1 | grep "^cfg5" | tail -1
timeout -k 10 200 python tools/quick_time.py cfg4 0.1 2 noise=1 $o 2>&1 | grep "^cfg4" | tail -1
done
