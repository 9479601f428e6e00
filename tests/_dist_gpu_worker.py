"""Worker for the GPU test of the band-sharded entry point under the real RCCL backend (one rank on the
one GPU of the test box; NBLS_FORCE_DIST_PATH=1 takes the sharded code path even for world size 1)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    import torch.distributed as td
    torch.cuda.set_device(0)
    td.init_process_group('nccl', device_id=torch.device('cuda', 0))
    os.environ['NBLS_FORCE_DIST_PATH'] = '1'
    from narrow_band_least_squares_amd import (narrow_band_least_squares, narrow_band_least_squares_parallel,
                                               synthetic)
    for name, alpha in (('cfg1', 1.0), ('cfg2', 0.5)):
        c = synthetic.build_config(name, 0.1)
        nb = 5
        fr = np.logspace(-2, 1, 40)
        w = np.zeros(40)
        args = (c['WINLEN_list'][:nb], 0.5, alpha, c['st'], None, None, nb, w, w, c['freqlist'][:nb + 1], c['band_type'],
                fr, 'butter', 2, 0.01)
        par = narrow_band_least_squares_parallel(*args, rij=c['rij'])
        ser = narrow_band_least_squares(*args, rij=c['rij'])
        assert par[6] == ser[6]
        for i in (0, 1, 2, 3, 5, 7, 8):
            np.testing.assert_array_equal(par[i], ser[i])
        if ser[4] is None:
            assert par[4] is None
        else:
            assert list(par[4].keys()) == list(ser[4].keys())
            for k in ser[4]:
                np.testing.assert_array_equal(par[4][k], ser[4][k])
    print('DIST_GPU_OK')
    td.destroy_process_group()


if __name__ == '__main__':
    main()
