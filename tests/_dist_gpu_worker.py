"""Worker for the GPU test of the sharded entry point on the library's own RCCL gather (no PyTorch).
Runs in its own process because a communicator, once made, lives as long as the process.

  mode 'all'   one process drives every visible GPU (this box: one): ncclCommInitAll + gather to root 0
  mode 'rank'  process-per-GPU form with a one-rank world: the 128-byte id, ncclCommInitRank, all-gather
  mode 'loopN' (N = 2, 3): N RANKS of one process on this box's one GPU — N library handles, N launch threads, N result
               blocks in HBM, nbls_comm_gather with root 0 — over the loopback stand-in of tests/c_caller/loopback_rccl.cpp
               (device-to-device copies where RCCL would use xGMI; the caller names it in NBLS_TEST_TRANSPORT — read HERE and handed
               to the library through nbls_comm_set_library, the library itself reads no environment — and sets NBLS_DEVICES=0,0[,0])
  mode 'procfail'  the same launcher form, with rank 1 failing before it can plan (status word, nobody hangs), then a healthy call
  mode 'proc'  one PROCESS per rank under a launcher (python -m torch.distributed.run --nproc-per-node 2 ...), all of them
               on this box's one GPU (NBLS_DEVICE=0): the id over the TCP side channel, ncclCommInitRank with a world of
               two, the all-gather — the stand-in moves the blocks through shared host memory
All must equal the serial call bit for bit, bands- and windows-sharded."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    mode = sys.argv[1]
    os.environ['NBLS_FORCE_DIST_PATH'] = '1'
    from narrow_band_least_squares_amd import (narrow_band_least_squares, narrow_band_least_squares_parallel,
                                               synthetic, dist, engine)
    assert 'torch' not in sys.modules
    if os.environ.get('NBLS_TEST_TRANSPORT'):
        dist.set_transport_library(os.environ['NBLS_TEST_TRANSPORT'], allow_shared_device=True)
    if mode == 'rank':
        # what dist.get_group() does for WORLD_SIZE > 1, with a world of one
        import ctypes as C
        h = engine.get_handle()
        uid = (C.c_char * 128)()
        assert h.lib.nbls_comm_unique_id(uid, 128) == 0
        h._chk(h.lib.nbls_comm_init_rank(h._h, bytes(uid), 1, 0))
        dist._group_override = dist.Group([h], [0], 1, root=-1)
    if mode == 'procfail':
        # rank 1 fails before it can plan: it must still enter the gather (an empty block, status word set), rank 0 must
        # learn of it from the status word, nobody may hang — and the communicator must serve a healthy call afterwards
        rank = int(os.environ['RANK'])
        c = synthetic.build_config('cfg2', 0.1)
        fr = np.logspace(-2, 1, 40)
        w = np.zeros(40)
        args = (c['WINLEN_list'][:5], 0.5, 0.5, c['st'], None, None, 5, w, w, c['freqlist'][:6], c['band_type'], fr, 'butter', 2, 0.01)
        real_prepare = engine.prepare
        if rank == 1:
            def broken(*a, **k):
                raise MemoryError('injected failure on rank 1')
            engine.prepare = broken
        try:
            narrow_band_least_squares_parallel(*args, rij=c['rij'])
            raise SystemExit('rank %d: the failed call returned' % rank)
        except MemoryError as e:
            assert rank == 1 and 'injected' in str(e)
        except RuntimeError as e:
            assert rank == 0 and 'rank(s) [1] failed' in str(e), str(e)
        engine.prepare = real_prepare
        par = narrow_band_least_squares_parallel(*args, rij=c['rij'])
        ser = narrow_band_least_squares(*args, rij=c['rij'])
        for i in (0, 1, 2, 3, 5, 7, 8):
            np.testing.assert_array_equal(par[i], ser[i])
        assert list(par[4].keys()) == list(ser[4].keys())
        sys.stdout.write('DIST_GPU_OK %s rank %d\n' % (mode, rank))
        sys.stdout.flush()
        return
    # (the fourth case: an HBM budget of two bands per pass — the share runs in three rounds, the block assembled on the
    #  host goes back through nbls_load_result_block and out through the same gather)
    # (stream: NBLS_STREAM_RESULTS — '1' = the dictionary from the ranks' streamed batches although the call is small, None = as
    #  a call of this size runs by itself, in one piece; rows: the passes are queued while their traces are still going up)
    for name, alpha, shard, per_pass, stream, rows_in_flight in (('cfg1', 1.0, 'bands', 0, None, False), ('cfg2', 0.5, 'bands', 0, '1', True),
                                                                 ('cfg2', 0.5, 'bands', 0, None, False), ('cfg2', 0.75, 'windows', 0, '1', True),
                                                                 ('cfg2', 0.5, 'bands', 2, None, False)):
        os.environ['NBLS_SHARD'] = shard
        os.environ.pop('NBLS_STREAM_RESULTS', None)
        if stream:
            os.environ['NBLS_STREAM_RESULTS'] = stream
        engine.ROW_PIPELINE_MIN_BYTES = 0 if rows_in_flight else 256 << 20
        c = synthetic.build_config(name, 0.1)
        nb = 9 if mode == 'loop8' else 5             # (eight ranks: some take two bands, most one)
        os.environ.pop('NBLS_MAX_FILTERED_GB', None)
        if per_pass:
            nchans, npts = c['data'].shape
            os.environ['NBLS_MAX_FILTERED_GB'] = repr((per_pass + 0.5) * 8.0 * nchans * (npts + 64) / 2.0 ** 30)
            assert engine.max_bands_per_pass(nchans, npts) == per_pass
        fr = np.logspace(-2, 1, 40)
        w = np.zeros(40)
        args = (c['WINLEN_list'][:nb], 0.5, alpha, c['st'], None, None, nb, w, w, c['freqlist'][:nb + 1], c['band_type'],
                fr, 'butter', 2, 0.01)
        par = narrow_band_least_squares_parallel(*args, rij=c['rij'])
        os.environ.pop('NBLS_MAX_FILTERED_GB', None)
        os.environ.pop('NBLS_STREAM_RESULTS', None)
        engine.ROW_PIPELINE_MIN_BYTES = 256 << 20
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
    g = dist.get_group()
    want_world = int(mode[4:]) if mode.startswith('loop') else 1
    if mode == 'proc':
        assert int(os.environ['WORLD_SIZE']) > 1 and g.world == int(os.environ['WORLD_SIZE']) and len(g.handles) == 1
        sys.stdout.write('DIST_GPU_OK %s rank %s\n' % (mode, os.environ['RANK']))      # (ONE write: two ranks share the pipe)
        sys.stdout.flush()
        return
    assert g is not None and g.world == want_world and len(g.handles) == want_world and g.handles[0].lib.nbls_version() >= 200
    assert len({id(h) for h in g.handles}) == want_world
    assert 'torch' not in sys.modules
    print('DIST_GPU_OK', mode)


if __name__ == '__main__':
    main()
