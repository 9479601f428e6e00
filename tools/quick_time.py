"""Stage timings of one named configuration on the GPU (developer tool)."""
import sys
import time

import numpy as np

sys.path.insert(0, __file__.rsplit('/', 2)[0])
from narrow_band_least_squares_amd import engine, synthetic, planner  # noqa: E402


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else 'cfg3'
    scale = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
    impl = int(sys.argv[3]) if len(sys.argv) > 3 else 0
    reps = int(sys.argv[4]) if len(sys.argv) > 4 else 3
    c = synthetic.build_config(name, scale=scale)
    data, fs, t0 = engine.stream_to_array(c['st'])
    step = 2 if c['band_type'] == '2_octave_over' else 1
    edges = [(c['freqlist'][i], c['freqlist'][i + step]) for i in range(c['NBANDS'])]
    h = engine.get_handle()
    h.set_profiling(True)
    for r in range(reps):
        t = time.time()
        res = engine.process(data, fs, t0, c['rij'], edges, c['WINLEN_list'], c['overlap'], c['alpha'],
                             c['ftype'], c['order'], c['ripple'], xcorr_impl=impl)
        wall = time.time() - t
        tm = h.timings()
        U = int(res.nwin.sum())
        print('%s scale=%g impl=%d units=%d wall=%.3fs filter=%.2fms xcorr=%.2fms solve=%.2fms total=%.2fms -> %.0f solves/s (device)'
              % (name, scale, impl, U, wall, tm['filter_ms'], tm['xcorr_ms'], tm['solve_ms'], tm['total_ms'],
                 U / (tm['total_ms'] * 1e-3)), flush=True)
    if impl == 3:
        print('screen stats (last batch):', h.screen_stats())
    import os
    if os.environ.get('NBLS_SCREEN_STAMPS') == '1':
        st = h.screen_stamps()
        tot = st['total'] or 1.0
        print('screen stamps (mean cycles per workgroup, last batch):',
              {k: '%.0f (%.0f%%)' % (v, 100 * v / tot) for k, v in st.items()})
    if os.environ.get('NBLS_LTS_STAMPS') == '1':
        st = h.lts_stamps()
        tot = st['total'] or 1.0
        print('lts stamps (mean cycles per wave):', {k: '%.0f (%.0f%%)' % (v, 100 * v / tot) for k, v in st.items()})
    n = int(res.nwin[len(edges) // 2])
    print('mid band: median baz %.3f vel %.4f mdccm %.3f' % (np.nanmedian(res.baz[len(edges) // 2, :n]),
          np.nanmedian(res.vel[len(edges) // 2, :n]), np.nanmedian(res.mdccm[len(edges) // 2, :n])))


if __name__ == '__main__':
    main()
