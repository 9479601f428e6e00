set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "variants or any_array_size or hostile or cfg1" 2>&1 | tail -2
for i in 1 2 3; do
timeout -k 10 120 python tools/quick_time.py cfg3 1 3 2>&1 | grep "^cfg3" | tail -1
done
timeout -k 10 120 python tools/quick_time.py cfg2 1 3 2>&1 | grep "^cfg2" | tail -1
