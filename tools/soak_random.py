"""Developer: one-off soak of the whole drop-in call against the oracle on further seeds of
tests/test_gpu_parity.py::test_random_configurations_against_oracle (also with pipelined band groups and gaps).

    python tools/soak_random.py FIRST LAST [groups]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'oracle'), os.path.join(ROOT, 'tests')]
import contextlib, io
import numpy as np
import nbls_oracle as oracle
import test_gpu_parity as T
first, last = int(sys.argv[1]), int(sys.argv[2])
if len(sys.argv) > 3:
    os.environ['NBLS_PIPELINE_GROUPS'] = sys.argv[3]
bad = 0
for seed in range(first, last):
    try:
        with contextlib.redirect_stdout(io.StringIO()):
            T.test_random_configurations_against_oracle.__wrapped__(oracle, seed) if hasattr(T.test_random_configurations_against_oracle, '__wrapped__') else T.test_random_configurations_against_oracle(oracle, seed)
    except Exception as e:      # noqa: BLE001
        bad += 1
        print('seed', seed, 'FAILED:', type(e).__name__, str(e)[:300], flush=True)
    if (seed - first) % 20 == 19:
        print('... up to seed', seed, 'failures so far', bad, flush=True)
print('soak done: seeds %d..%d, %d failures' % (first, last - 1, bad))
