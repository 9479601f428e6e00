set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 800 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "variants or filter or cfg1 or product_against or segment" > gpurun_out/t1.log 2>&1 || { tail -30 gpurun_out/t1.log; exit 1; }
tail -3 gpurun_out/t1.log
for i in 1 2; do
timeout -k 10 120 python tools/quick_time.py cfg3 1 3 2>&1 | grep "^cfg3" | tail -1
timeout -k 10 120 python tools/quick_time.py cfg3 1 3 filter_store_y1=1 2>&1 | grep "^cfg3" | tail -1
done
timeout -k 10 120 python tools/quick_time.py cfg4 0.1 2 2>&1 | grep "^cfg4" | tail -1
timeout -k 10 120 python tools/quick_time.py cfg4 0.1 2 filter_store_y1=1 2>&1 | grep "^cfg4" | tail -1
