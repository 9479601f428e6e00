"""Stage timings of one named configuration on the GPU (developer tool): all bands as ONE pass, planned once,
executed back to back.

    python tools/quick_time.py cfg3 [scale] [reps] [key=value ...]      # key=value -> Handle.set_option
    NBLS_LIB=narrow_band_least_squares_amd/csrc/libnbls_hip_dev.so python tools/quick_time.py cfg3 1 3 screen_stamps=1

With the developer build (`make -C narrow_band_least_squares_amd/csrc dev`) the options screen_stamps / lts_stamps
print the in-kernel phase stamps, and ablate=<bits> skips kernel parts (results wrong, timing only)."""
import sys

import numpy as np

sys.path.insert(0, __file__.rsplit('/', 2)[0])
from narrow_band_least_squares_amd import engine, synthetic  # noqa: E402


def main():
    pos = [a for a in sys.argv[1:] if '=' not in a]
    opts = dict(a.split('=') for a in sys.argv[1:] if '=' in a)
    name = pos[0] if pos else 'cfg3'
    scale = float(pos[1]) if len(pos) > 1 else 1.0
    reps = int(pos[2]) if len(pos) > 2 else 3
    impl = int(opts.pop('impl', 0))
    nbands = int(opts.pop('bands', 0))
    noise = int(opts.pop('noise', 0))
    gap_ms = float(opts.pop('gap_ms', 0))        # idle time between the passes (clock ramp-down experiment)
    stream = int(opts.pop('stream', 0))          # stream=1: the pass as a whole call runs it (per-batch solves, rows streamed to the host)
    c = synthetic.build_config(name, scale=scale)
    rows, fs, t0 = engine.stream_rows(c['st'])
    if noise:          # incoherent white noise of the same shape (no common signal)
        rng = np.random.default_rng(7)
        rows = [rng.standard_normal(len(r)) for r in rows]
    step = 2 if c['band_type'] == '2_octave_over' else 1
    nb = nbands or c['NBANDS']
    edges = [(c['freqlist'][i], c['freqlist'][i + step]) for i in range(nb)]
    h = engine.get_handle()
    for k, v in opts.items():
        h.set_option(k, int(v))
    h.set_profiling(True)
    prep = engine.prepare(len(rows), len(rows[0]), fs, c['rij'], edges, c['WINLEN_list'][:nb], c['overlap'], c['alpha'],
                          c['ftype'], c['order'], c['ripple'])
    engine.launch(h, rows, prep, xcorr_impl=impl, stream=bool(stream))
    h.sync()
    U = int(prep.nwin.sum())
    import time
    for r in range(reps):
        if gap_ms:
            time.sleep(gap_ms * 1e-3)
        h.execute()
        h.sync()
        tm = h.timings()
        print('%s scale=%g units=%d filter=%.2f quantize=%.2f screen=%.2f verify=%.2f solve=%.2f total=%.2f ms -> %.0f solves/s (device)'
              % (name, scale, U, tm['filter_ms'], tm['quantize_ms'], tm['screen_ms'], tm['verify_ms'], tm['solve_ms'],
                 tm['total_ms'], U / (tm['total_ms'] * 1e-3)), flush=True)
    if tm['xcorr_impl'] == 3:
        print('screen stats (last batch):', h.screen_stats())
    if opts.get('screen_stamps'):
        st = h.screen_stamps()
        tot = st['total'] or 1.0
        print('screen stamps (mean cycles per workgroup, last batch):',
              {k: '%.0f (%.0f%%)' % (v, 100 * v / tot) for k, v in st.items()})
    if opts.get('lts_stamps'):
        st = h.lts_stamps()
        tot = st['total'] or 1.0
        print('lts stamps (mean cycles per wave):', {k: '%.0f (%.0f%%)' % (v, 100 * v / tot) for k, v in st.items()})
        if len(rows) > 8:
            print('large-array LTS kernel, C-step phase (thread 0 cycles):', {k: '%.0f' % v for k, v in h.lts_coop_breakdown().items()})
    out = h.fetch_packed()
    mid = nb // 2
    n = int(prep.nwin[mid])
    print('mid band: median baz %.3f vel %.4f mdccm %.3f' % (np.nanmedian(out['baz'][mid, :n]), np.nanmedian(out['vel'][mid, :n]),
                                                            np.nanmedian(out['mdccm'][mid, :n])))


if __name__ == '__main__':
    main()
