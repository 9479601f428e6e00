"""CPU rehearsal of the N>1 host logic of narrow_band_least_squares_parallel(): band / window
partition, equal-sized result blocks, the status word, reassembly and the stdict merge.

There is no GPU here, so the two library calls of that path are replaced by stand-ins:
``engine.launch`` computes a rank's share with the CPU oracle and leaves a result block laid out
exactly as the GPU writes it (nbls_result_layout), and the Group's ``gather`` moves the blocks with a
gloo all-gather (process-per-GPU form, this file run under torch.distributed.run with 2 ranks) or by
plain concatenation (one-process form, ``run_single_process`` imported by tests/test_host.py).
The RCCL gather itself is exercised on the GPU box (tests/test_gpu_parity.py)."""
import os
import time
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'oracle'))
import nbls_oracle as oracle  # noqa: E402
from narrow_band_least_squares_amd import dist, engine  # noqa: E402
import importlib  # noqa: E402
nbls_mod = importlib.import_module('narrow_band_least_squares_amd.narrow_band_least_squares')


class FakeHandle:
    """CPU stand-in of _hip.Handle for the host logic of the sharded call: records the declared trace shape and the
    rows its upload thread delivered."""
    block = None
    shape = None
    uploaded = None

    def set_trace_shape(self, nchans, npts, fs):
        self.shape = (int(nchans), int(npts), float(fs))
        self.uploaded = None

    def upload_rows(self, rows):
        assert self.shape is not None and len(rows) == self.shape[0] and all(len(r) == self.shape[1] for r in rows)
        self.uploaded = [np.array(r) for r in rows]

    # several HBM rounds of one rank's share: every round's block is fetched, the assembled one loaded back
    rounds = 0

    def fetch_packed(self):
        nb, VL, MB = self.last_shape
        g = np.frombuffer(self.block[:4 * nb * VL * 8].tobytes(), dtype=np.float64).reshape(4, nb, VL)
        m = self.block[4 * nb * VL * 8:].reshape(nb, VL, MB)
        self.rounds += 1
        return dict(vel=g[0], baz=g[1], mdccm=g[2], sigma_tau=g[3], mask=m)

    def load_result_block(self, block):
        self.block = np.array(block, dtype=np.uint8)
        self.loaded = True

    # streamed passes (contiguous band shares): the block arrives in batches that cut through bands
    streamed = False
    waited = 0

    def _batches(self):
        U = sum(self.nwin_share)
        cuts = sorted({0, U} | {int(0.37 * U), int(0.8 * U)})
        return list(zip(cuts[:-1], cuts[1:]))

    def result_batches(self):
        return len(self._batches()) if self.streamed else 0

    def wait_result_batch(self, k):
        assert self.streamed
        nb, VL, MB = self.last_shape
        u0, u1 = self._batches()[k]

        def cell(u):
            off = 0
            for b, n in enumerate(self.nwin_share):
                if u < off + n:
                    return b * VL + (u - off)
                off += n
            raise AssertionError(u)
        c0, c1 = (cell(u0), cell(u1 - 1) + 1) if u1 > u0 else (0, 0)
        cells = nb * VL
        g = np.full((4, cells), np.nan)                                   # what no batch has delivered yet is undefined
        m = np.full((cells, MB), 0xA5, dtype=np.uint8)
        g[:, c0:c1] = np.frombuffer(self.block[:32 * cells].tobytes(), dtype=np.float64).reshape(4, cells)[:, c0:c1]
        m[c0:c1] = self.block[32 * cells:].reshape(cells, MB)[c0:c1]
        self.waited += 1
        return u0, u1, c0, c1, g, m


def make_fake_launch(gold, fail_rank=None, rank_of=None):
    """engine.launch stand-in: the oracle computes the share, the block is left on the fake handle."""
    edges_all = nbls_mod._band_edges(list(gold['freqlist']), str(gold['band_type']), range(len(gold['num_compute'])))

    def fake_launch(h, data, prep, bands=None, upload=True, window_slice=None, xcorr_impl=0, reserve_bytes=0,
                    trace_from=None, trace_ready=False, after=None, before_execute=None, stream=False, uncert=False):
        if fail_rank is not None and rank_of(h) == fail_rank:
            raise ValueError('injected failure on rank %d' % fail_rank)
        if before_execute is not None:       # the real launch joins the upload thread between plan and execute
            before_execute()
        if trace_ready:                      # the pass works on what the upload thread delivers: queued while the rows may
            t0 = time.time()                 # still be going up, the library waits for them itself (nbls_execute: row events)
            while h.uploaded is None:
                assert time.time() - t0 < 30, 'the trace never landed'
                time.sleep(0.001)
            data = h.uploaded
        idx = list(range(prep.nbands)) if bands is None else list(bands)
        nb, VL, MB, P = len(idx), prep.vector_len, prep.mask_bytes, prep.npairs
        grids = np.zeros((4, nb, VL))
        wts = np.zeros((nb, VL, P), dtype=np.uint8)
        arr = np.array(data)
        for n, b in enumerate(idx):
            st = oracle.make_stream(arr, prep.fs)
            stf, _, _ = oracle.filter_data(st, str(gold['ftype']), edges_all[b][0], edges_all[b][1], 2, 0.01)
            out, internals = oracle.ltsva(stf, None, None, float(gold['winlens'][b]), 0.5, prep.alpha, rij=gold['rij'],
                                          return_internals=True)
            k = len(out[0])
            lo, hi = 0, k
            if window_slice is not None:      # only this rank's slice of the rows, like the device path
                r, nsl = window_slice
                lo, hi = (k * r) // nsl, (k * (r + 1)) // nsl
            grids[0, n, lo:hi], grids[1, n, lo:hi] = out[0][lo:hi], out[1][lo:hi]
            grids[2, n, lo:hi], grids[3, n, lo:hi] = out[3][lo:hi], out[5][lo:hi]
            wts[n, lo:hi] = internals['weights'].T[lo:hi]
        mask = np.packbits(wts, axis=-1, bitorder='little')
        assert mask.shape[-1] == MB
        h.block = np.frombuffer(grids.tobytes() + mask.tobytes(), dtype=np.uint8)
        h.last_shape = (nb, VL, MB)
        h.streamed = bool(stream)
        h.nwin_share = [int(prep.nwin[b]) for b in idx]
        assert reserve_bytes >= len(h.block) + 8
    return fake_launch


def pad_block(h, block_bytes, status):
    blk = np.zeros(block_bytes, dtype=np.uint8)
    if h.block is not None:
        blk[:len(h.block)] = h.block
    blk[-8:] = np.array([status], dtype=np.int64).view(np.uint8)
    return blk


class LocalGroup:
    """One process, ``world`` fake devices, gather to root 0."""

    def __init__(self, world):
        self.handles = [FakeHandle() for _ in range(world)]
        self.ranks = list(range(world))
        self.world, self.root = world, 0

    def gather(self, block_bytes, status=0):
        return np.stack([pad_block(h, block_bytes, status) for h in self.handles])


class GlooGroup:
    """One process per rank, all-gather over gloo."""

    def __init__(self, rank, world):
        self.handles, self.ranks, self.world, self.root = [FakeHandle()], [rank], world, -1

    def gather(self, block_bytes, status=0):
        import torch
        import torch.distributed as td
        t = torch.from_numpy(pad_block(self.handles[0], block_bytes, status))
        outs = [torch.empty_like(t) for _ in range(self.world)]
        td.all_gather(outs, t)
        return np.stack([o.numpy() for o in outs])


def call_and_compare(gold, group, expect_failure=False):
    st = oracle.make_stream(gold['data'], float(gold['fs']), starttime=17884.0729166667)
    nb = len(gold['num_compute'])
    fr = gold['freq_resp']
    w = np.zeros(len(fr))
    args = (list(gold['winlens']), 0.5, float(gold['alpha']), st, None, None, nb, w, w, list(gold['freqlist']),
            str(gold['band_type']), fr, str(gold['ftype']), 2, 0.01)
    dist._group_override = group
    try:
        if expect_failure:
            try:
                nbls_mod.narrow_band_least_squares_parallel(*args, rij=gold['rij'])
            except (ValueError, RuntimeError):
                return nb
            raise AssertionError('the injected failure did not surface on this rank')
        got = nbls_mod.narrow_band_least_squares_parallel(*args, rij=gold['rij'])
    finally:
        dist._group_override = None
    exp = oracle.narrow_band_least_squares(*args, rij=gold['rij'])
    assert got[6] == exp[6], (got[6], exp[6])
    for i in (0, 1, 2, 3, 5, 7, 8):
        np.testing.assert_array_equal(got[i], exp[i])
    if exp[4] is None:
        assert got[4] is None
    else:
        assert list(got[4].keys()) == list(exp[4].keys())
        for k in exp[4]:
            np.testing.assert_array_equal(got[4][k], exp[4][k])
    return nb


def run_single_process(gold_name, mode, world, monkeypatch, bands_per_pass=None):
    """``bands_per_pass``: HBM budget (NBLS_MAX_FILTERED_GB) small enough for that many bands per pass, so that the
    ranks run their shares in several rounds and load the assembled block back before the gather."""
    gold = np.load(os.path.join(ROOT, 'tests', 'golden', gold_name + '.npz'), allow_pickle=False)
    group = LocalGroup(world)
    monkeypatch.setattr(engine, 'launch', make_fake_launch(gold))
    if mode == 'windows':
        monkeypatch.setenv('NBLS_SHARD', 'windows')
    if bands_per_pass:
        nchans, npts = gold['data'].shape
        monkeypatch.setenv('NBLS_MAX_FILTERED_GB', repr((bands_per_pass + 0.5) * 8.0 * nchans * (npts + 64) / 2.0 ** 30))
        assert engine.max_bands_per_pass(nchans, npts) == bands_per_pass
    nb = call_and_compare(gold, group)
    if bands_per_pass:
        assert any(getattr(h, 'loaded', False) and h.rounds >= 2 for h in group.handles), 'no rank needed several rounds'
    elif mode == 'bands' and float(gold['alpha']) < 1.0 and os.environ.get('NBLS_STREAM_RESULTS') == '1':
        # contiguous band shares under LTS: the dictionary was built from the ranks' streamed batches, in rank order
        assert all(h.streamed and h.waited == h.result_batches() for h in group.handles if h.block is not None), 'not streamed'
    else:
        assert not any(h.waited for h in group.handles)
    return nb


def main():
    import torch.distributed as td
    td.init_process_group('gloo')
    rank, world = td.get_rank(), td.get_world_size()
    mode = sys.argv[2] if len(sys.argv) > 2 else 'bands'
    if mode == 'windows':
        os.environ['NBLS_SHARD'] = 'windows'
    gold = np.load(os.path.join(ROOT, 'tests', 'golden', sys.argv[1] + '.npz'), allow_pickle=False)
    engine.launch = make_fake_launch(gold, fail_rank=1 if mode == 'fail' else None, rank_of=lambda h: rank)
    group = GlooGroup(rank, world)
    if mode == 'fail_early' and rank == 1:
        # this rank fails BEFORE it can plan (its trace upload cannot even be declared): it must still take part in
        # the gather, with its status word set, and both ranks must raise
        def boom(*a):
            raise RuntimeError('injected failure before planning on rank 1')
        group.handles[0].set_trace_shape = boom
    nb = call_and_compare(gold, group, expect_failure=mode in ('fail', 'fail_early'))
    if mode == 'bands' and float(gold['alpha']) < 1.0 and os.environ.get('NBLS_STREAM_RESULTS') == '1':
        # under a launcher too, a rank builds the dictionary entries of ITS bands from its streamed batches (rank 0 straight
        # into the result, rank 1 into a part merged behind rank 0's gathered entries); the key order is checked above
        hd = group.handles[0]
        assert hd.block is None or (hd.streamed and hd.waited == hd.result_batches() > 0), 'rank %d did not stream' % rank
    td.barrier()
    if rank == 0:
        print('DIST_OK world=%d bands=%d mode=%s' % (world, nb, mode))
    td.destroy_process_group()


if __name__ == '__main__':
    main()
