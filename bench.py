#!/usr/bin/env python3
"""Headline benchmark: (window x band) LTS solves/s.

    python bench.py --gpus N --steps K --warmup W [--config cfg3|cfg2|cfg4|cfg5] [--shard bands|traces]

A step is ONE WHOLE drop-in call — ``narrow_band_least_squares(...)`` exactly as example.py makes it
(reference narrow_band_least_squares.py:8-127; SURVEY.md §8(d): "wall time of the whole call including
H2D/D2H and the final gather"): upload of the raw trace from the stream's own buffers, filter design and
plan on the host, the device pass (filter + taper, pairwise cross-correlation / lag pick, MdCCM, FAST-LTS +
reweighting for every (band, window) unit), ONE D2H copy of the result block (grids + packed LTS weights),
the filter responses, and the reference's dropped-element dictionary ``stdict`` with one entry per window.
``value`` = units / that wall time.  Beside it: ``kernel_only_ms`` (the device pass alone — all bands as one pass,
executed back to back, HIP events on the library's stream), ``value_trace_resident`` (the same whole calls with the
samples already in HBM when the clock starts: ``with resident_trace(st)``, measured), and ``stage_ms``.  The filter-design
cache of the host planner is cleared before every step, so no step reuses host work of an earlier one.

N = 1: the configuration named by --config (default cfg-3 = BASELINE.json's 8-element metric config; cfg-4 is
run as ONE GPU's 12-band share of the 96-band / 24 h job, see --band-share).
N > 1: --shard bands (default) is ONE call of ``narrow_band_least_squares_parallel`` sharded by bands over the
GPUs with the single RCCL gather inside the library (strong scaling: total work fixed) — either one process per GPU
(started by ``python -m torch.distributed.run``, WORLD_SIZE = N) or, WITHOUT a launcher, ``python bench.py --gpus N``:
one process that drives GPUs 0..N-1 (``nbls_comm_init_all``).  Fewer than N GPUs visible, or a communicator that
cannot be formed: one JSON line with ``value`` null and ``status: failed``, exit code 2 — never a silent 1-GPU or
weak-scaling number under the same metric.  --shard traces (launcher only) gives every rank its own independent
trace and whole call (weak scaling, no collective).  torch.distributed (gloo) is used by THIS SCRIPT only for
the barrier around the timed region and the max-over-ranks — the product path has no PyTorch in it.

The JSON line carries ``roofline`` (dominant kernel = the int8-MFMA screening correlator; duration measured
live with HIP events on the library's stream), ``roofline_hbm`` (the mandated HBM figure: algorithmic bytes /
wall time), ``noise`` (the same call on an incoherent-noise trace: the screening work is data dependent),
``cpu_baseline`` (the CPU oracle timed on this box's host cores: 1 core and all cores) and ``env``.
"""
import argparse
import contextlib
import json
import math
import os
import platform
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from narrow_band_least_squares_amd import (dist, engine, planner, synthetic,  # noqa: E402
                                           narrow_band_least_squares, narrow_band_least_squares_parallel)

FP64_MFMA_PEAK_TFLOPS = 78.6     # MI355X FP64 matrix = vector peak (AMD datasheet; the guide lists none)
I8_MFMA_PEAK_TOPS = 5000.0       # MI355X_MICROARCH.md: int8 MFMA = 2x the bf16 dense rate (~2.5 PF) per clock
HBM_PEAK_GBS = 8000.0            # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)

LABELS = {
    'cfg2': 'cfg-2: 6-element synthetic plane wave, 24 log bands 0.1-5 Hz, LTS alpha=0.75, 1 h @ 20 Hz',
    'cfg3': 'cfg-3: 8-element synthetic plane wave, 48 log bands 0.1-10 Hz, LTS alpha=0.5, 6 h @ 40 Hz',
    'cfg4': 'cfg-4: 16-element synthetic plane wave, 96 log bands 0.1-10 Hz, LTS alpha=0.5 (500 LCG starts), 24 h @ 100 Hz',
    'cfg5': 'cfg-5: 32-element synthetic plane wave, 128 log bands 0.1-5 Hz, LTS alpha=0.5 (500 LCG starts), 1 h @ 20 Hz',
    'cfg1': 'cfg-1: 6-element synthetic plane wave, 10 linear bands 0.5-5 Hz, OLS, 20 min @ 20 Hz',
    'cfg1b': 'cfg-1b: example.py parameters, 8-element synthetic, 8 log bands, cheby1, adaptive windows, OLS',
}


def cpu_model():
    try:
        for line in open('/proc/cpuinfo'):
            if line.startswith('model name'):
                return line.split(':', 1)[1].strip()
    except OSError:
        pass
    return platform.processor() or 'unknown'


def cpu_task(args):
    """One band of the CPU oracle (the unit of the reference's joblib parallelism)."""
    sys.path.insert(0, os.path.join(ROOT, 'oracle'))
    import nbls_oracle as o
    data, fs, rij, fmin, fmax, winlen, alpha, ftype, order, ripple = args
    st = o.make_stream(data, fs)
    stf, _, _ = o.filter_data(st, ftype, fmin, fmax, order, ripple)
    out = o.ltsva(stf, None, None, winlen, 0.5, alpha, rij=rij)
    return len(out[0])


def cpu_baseline(c, edges, winlens, budget_s=8.0):
    """The oracle ("port": lts_array's source is absent, so the reference's example_parallel.py path cannot
    run here) timed on the host cores on a bounded sample of the same workload: first on ONE core (one
    band), then one band per core through joblib — the reference's parallel structure
    (narrow_band_least_squares.py:285).  The sample is sized from a probe so each leg takes ~budget_s."""
    from joblib import Parallel, delayed
    cores = os.cpu_count() or 1
    nb = min(cores, len(edges))
    pick = [int(round(i * (len(edges) - 1) / max(1, nb - 1))) for i in range(nb)] if nb > 1 else [len(edges) // 2]
    fs = c['fs']
    data = np.array(c['data'])

    def task(b, npts):
        return (data[:, :npts], fs, c['rij'], edges[b][0], edges[b][1], winlens[b], c['alpha'], c['ftype'], c['order'],
                c['ripple'])
    wl = max(winlens)
    probe_s = 5.0 * wl
    mid = pick[len(pick) // 2]
    t = time.time()
    n = cpu_task(task(mid, int(probe_s * fs)))
    per_unit = (time.time() - t) / max(1, n)
    units_per_band = max(8, int(budget_s / per_unit))
    seconds = min(c['dur'], (units_per_band + 1) * wl * 0.5 + wl)
    npts = int(seconds * fs)
    t = time.time()
    n1 = cpu_task(task(mid, npts))
    wall1 = time.time() - t
    t = time.time()
    counts = Parallel(n_jobs=nb)(delayed(cpu_task)(task(b, npts)) for b in pick)
    wall = time.time() - t
    return {'value': sum(counts) / wall, 'unit': 'solves/s', 'cores': nb, 'kind': 'port',
            'one_core_value': n1 / wall1, 'cpu_model': cpu_model(), 'host_cores': cores,
            'sample': '%d bands x first %.0f s of the trace (%d units; 1-core leg: 1 band, %d units), NumPy/SciPy oracle, '
                      'joblib one band per core' % (nb, seconds, sum(counts), n1)}


def ltsva_entry(args):
    """The step either side of the hot path: the broadband call of example.py:108-109,
    ``stf, Fs, sos = filter_data(st, FILTER_TYPE, FMIN, FMAX, ...)`` then ``ltsva(stf, lat, lon, WINLEN, WINOVER, ALPHA)``
    with WINLEN = 50 s (example.py:60) over the whole band of the configuration, all eight returns (the last two — the
    confidence intervals — computed on the GPU behind the solve).  A step = both calls; the two legs are also timed apart."""
    import io
    from narrow_band_least_squares_amd import filter_data, ltsva
    c = synthetic.build_config(args.config, scale=args.scale)
    st, rij, fs = c['st'], c['rij'], c['fs']
    fmin, fmax, winlen = c['freqlist'][0], c['freqlist'][-1], 50.0
    h = engine.get_handle()

    def one():
        t0 = time.perf_counter()
        stf, _, _ = filter_data(st, c['ftype'], fmin, fmax, c['order'], c['ripple'])
        t1 = time.perf_counter()
        with contextlib.redirect_stdout(io.StringIO()):
            out = ltsva(stf, None, None, winlen, c['overlap'], c['alpha'], False, rij=rij)
        return out, (t1 - t0) * 1e3, (time.perf_counter() - t1) * 1e3
    for _ in range(args.warmup):
        one()
    h.sync()
    tf, tl, held = [], [], []
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out, a_, b_ = one()
        tf.append(a_)
        tl.append(b_)
        held.append(out)
    h.sync()
    el = time.perf_counter() - t0
    n = len(out[0])
    line = {'metric': '(window x band) LTS solves/sec, 8-element synthetic', 'entry': 'ltsva (broadband filter_data -> ltsva, example.py:108-109)',
            'value': n * args.steps / el, 'unit': 'solves/s', 'n_gpus': 1, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': el / args.steps * 1e3, 'filter_data_ms_median': float(np.median(tf)), 'ltsva_ms_median': float(np.median(tl)),
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f64', 'data': 'synthetic',
            'config': {'workload': LABELS.get(args.config, args.config) + ' — ONE band %.3g-%.3g Hz, %s order %d, %g s windows %d%% overlap, alpha %g'
                                   % (fmin, fmax, c['ftype'], c['order'], winlen, int(round(c['overlap'] * 100)), c['alpha']),
                       'step': 'filter_data (upload, GPU filter + taper, the filtered stream back on the host) then ltsva (upload of the filtered '
                               'stream, correlation, solve, confidence intervals on the GPU, dictionary)',
                       'units_per_call': n, 'elements': len(st), 'window_samples': int(winlen * fs), 'scale': args.scale},
            'uncertainty': {'vel_uncert_median_km_s': float(np.nanmedian(out[6])), 'baz_uncert_median_deg': float(np.nanmedian(out[7])),
                            'where': 'GPU: uncertainty_kernel behind the solve (csrc/solve.hip); no host loop over windows'},
            'roofline': None, 'cpu_baseline': None,
            'note': 'a one-band call: PCIe-bound (the trace goes up twice and the filtered stream comes down once: 3 x 8 N npts bytes); '
                    'roofline / cpu_baseline ride on the narrow-band line (--entry nbls)'}
    print(json.dumps(line), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--config', default='cfg3')
    ap.add_argument('--shard', default='bands', choices=['bands', 'traces'],
                    help='N > 1: one call sharded by bands + RCCL gather (strong scaling) | one trace per rank (weak)')
    ap.add_argument('--band-share', default=None,
                    help='"k/n": run only the k-th of n band shares of the config (default 0/8 for cfg4, all bands otherwise)')
    ap.add_argument('--scale', type=float, default=1.0, help='shorten the trace (debug only; invalidates the metric)')
    ap.add_argument('--independent-calls', choices=['auto', 'on', 'off'], default='auto',
                    help='after the main measurement, time one whole single-GPU call per rank on the rank\'s own trace '
                         '(no collective; extra field `independent_calls`).  auto: in band-sharded runs under a launcher')
    ap.add_argument('--entry', default='nbls', choices=['nbls', 'ltsva'],
                    help='nbls (default): the narrow-band call of the metric.  ltsva: the broadband route of example.py:108-109 — '
                         'filter_data(st, ...) then ltsva(stf, ...) over the configuration\'s trace (one band, WINLEN = 50 s), '
                         'with the confidence intervals; its own JSON line, never BENCH\'s `value` of the narrow-band metric')
    ap.add_argument('--transport-lib', default=None,
                    help='REHEARSAL ONLY (one-GPU box): a library with RCCL\'s ten entry points to use instead of RCCL, several '
                         'ranks on one device (tests/c_caller/loopback_rccl.cpp); the numbers of such a run mean nothing')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-noise', action='store_true')
    args = ap.parse_args()

    if args.transport_lib:
        dist.set_transport_library(args.transport_lib, allow_shared_device=True)
    world = int(os.environ.get('WORLD_SIZE', '1'))       # PROCESSES of this job (a launcher's ranks); 1 without a launcher
    rank = int(os.environ.get('RANK', '0'))
    # Two ways to use N GPUs (dist.py): one process per GPU under a launcher (WORLD_SIZE = N), or — no launcher —
    # ONE process that drives GPUs 0..N-1 (nbls_comm_init_all, a handle per device): `python bench.py --gpus N`.
    one_process = world == 1 and args.gpus > 1
    ngpu = world if world > 1 else max(1, args.gpus)
    multi = ngpu > 1
    td = None
    if world > 1:
        import torch.distributed as td          # barrier + max-over-ranks of this script only
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        td.init_process_group('gloo')
    if args.gpus != world and rank == 0 and world > 1:
        print('warning: --gpus %d but WORLD_SIZE %d' % (args.gpus, world), file=sys.stderr)

    def give_up(note, code=2):
        """A run that cannot measure what was asked says so in ONE JSON line (value null) and exits non-zero."""
        if rank == 0:
            print(json.dumps({'metric': '(window x band) LTS solves/sec, 8-element synthetic', 'value': None, 'unit': 'solves/s',
                              'n_gpus': ngpu, 'steps': args.steps, 'warmup': args.warmup, 'status': 'failed', 'note': note}), flush=True)
        if td is not None:
            with contextlib.suppress(Exception):
                td.destroy_process_group()
        sys.exit(code)

    if one_process:
        if args.shard == 'traces':
            give_up('--shard traces needs one process per GPU (python -m torch.distributed.run --nproc-per-node %d bench.py ...)' % ngpu)
        if os.environ.get('NBLS_DEVICES'):
            ndev = len([x for x in os.environ['NBLS_DEVICES'].split(',') if x.strip() != ''])
        else:
            ndev = len(dist.visible_devices())
        if ndev < ngpu:
            give_up('--gpus %d asked, %d GPU(s) visible to this process: not measured' % (ngpu, ndev))
        os.environ.setdefault('NBLS_DEVICES', ','.join(str(d) for d in range(ngpu)))
        os.environ['NBLS_DEVICES'] = ','.join(os.environ['NBLS_DEVICES'].split(',')[:ngpu])
    shard_traces = world > 1 and args.shard == 'traces'
    if args.entry == 'ltsva':
        if ngpu != 1:
            give_up('--entry ltsva is a single-GPU measurement')
        return ltsva_entry(args)

    seed = synthetic.SEED + 1 + (rank if shard_traces else 0)
    c = synthetic.build_config(args.config, scale=args.scale, trace_seed=seed)
    nb_all = c['NBANDS']
    share = args.band_share or ('0/8' if args.config == 'cfg4' and not multi else None)
    bands = list(range(nb_all))
    if share:
        k, n = (int(x) for x in share.split('/'))
        npairs = c['N'] * (c['N'] - 1) // 2
        costs = dist.band_costs(c['npts'], c['fs'], list(c['WINLEN_list']), c['overlap'], npairs)
        bands = dist.shard_bands(costs, n)[k]
    freqlist = [c['freqlist'][b] for b in bands] + [c['freqlist'][bands[-1] + 1]]
    contiguous = bands == list(range(bands[0], bands[-1] + 1))
    winlens = [c['WINLEN_list'][b] for b in bands]
    nb = len(bands)
    st, rij, fs = c['st'], c['rij'], c['fs']
    nchans, npts = len(st), len(st[0].data)
    fr = np.logspace(np.log10(0.01), np.log10(fs / 2), 1000)       # example.py:117-119
    wdummy = np.zeros(len(fr))

    if contiguous:
        call_args = (winlens, c['overlap'], c['alpha'], st, None, None, nb, wdummy, wdummy, freqlist, c['band_type'], fr,
                     c['ftype'], c['order'], c['ripple'])
    else:
        # a non-contiguous band share cannot be expressed through freqlist: give every band its own edges
        # by calling with a 'per-band' list (NBANDS bands, edges b, b+1 of an interleaved list) — instead
        # use the engine-level entry with explicit edges
        call_args = None
    edges = [(c['freqlist'][b], c['freqlist'][b + 1]) for b in bands]

    # NBLS_DEVICE, else LOCAL_RANK (one process per GPU), else 0.  A rank without a GPU (fewer devices than ranks) must not
    # leave the others waiting in a collective: every rank learns of it and the run ends with the "failed" line.
    h, herr = None, None
    try:
        h = engine.get_handle()
    except Exception as e:      # noqa: BLE001
        herr = 'rank %d cannot open its GPU (%s: %s)' % (rank, type(e).__name__, e)
    if td is not None:
        import torch
        flag = torch.tensor([1 if herr else 0])
        td.all_reduce(flag, op=td.ReduceOp.MAX)
        if int(flag.item()) and herr is None:
            herr = 'another rank cannot open its GPU (fewer GPUs than ranks?)'
    if herr:
        print(herr, file=sys.stderr)
        give_up(herr + '; not measured')

    def one_call(stream, resident=False):
        planner.design_cache_clear()
        # the reference's "CAUTION: BT < 5!" prints (narrow_band_least_squares.py:86-87) still happen, but go to
        # stderr so that stdout carries the one JSON line only
        with contextlib.redirect_stdout(sys.stderr):
            if multi and not shard_traces:
                return narrow_band_least_squares_parallel(*call_args[:3], stream, *call_args[4:], rij=rij)
            if call_args is not None:
                return narrow_band_least_squares(*call_args[:3], stream, *call_args[4:], rij=rij)
            return share_call(stream, resident)

    def share_call(stream, resident):
        """The same whole call for a non-contiguous band share: the body of narrow_band_least_squares() with
        explicit band edges (a share of the bands cannot be expressed through ``freqlist``)."""
        from narrow_band_least_squares_amd.narrow_band_least_squares import _band_prefix, _vector_len
        from scipy import signal
        rows, fs_, t0 = engine.stream_rows(stream)
        vl = _vector_len(winlens, c['overlap'], stream)
        w_rows = np.zeros((nb, len(fr)), dtype=complex)
        h_rows = np.zeros((nb, len(fr)), dtype=complex)
        lts_ = c['alpha'] < 1.0

        def host_side(res):
            fast = planner.sosfreqz_bands(res.sos, fr, fs_)
            for n_, s_ in enumerate(res.sos):
                w_rows[n_], h_rows[n_] = signal.sosfreqz(s_, fr, fs=fs_) if fast is None else (fast[0], fast[1][n_])
            if lts_:
                res.keys = engine.time_key_text(res.t, res.nwin, [_band_prefix(b + 1) for b in bands])
                res.stdict = engine.new_stdict(engine.n_keys(res.keys))
                res.pattern_cache = engine.new_pattern_cache()

        def group_done(res, b0, b1):
            if lts_:
                engine.stdict_from_mask(res.mask[b0:b1], res.nwin[b0:b1], res.pair_idx, res.nchans, res.keys,
                                        into=res.stdict, k0=int(np.sum(res.nwin[:b0])), cache=res.pattern_cache)
        res = engine.process(rows, fs_, t0, rij, edges, winlens, c['overlap'], c['alpha'], c['ftype'], c['order'],
                             c['ripple'], vector_len=vl, host_overlap=host_side, group_done=group_done)
        return (res.vel, res.baz, res.mdccm, res.t, res.stdict if lts_ else None, res.sigma_tau, [int(x) for x in res.nwin],
                w_rows, h_rows)

    def barrier():
        h.sync()
        if one_process:
            for hd in dist.get_group().handles:
                hd.sync()
        if td is not None:
            td.barrier()

    def measure_kernels(stream, steps):
        """The device pass alone: all bands of this rank as ONE pass, planned once, executed back to back with
        HIP-event timing of the stages on the library's stream (what r01's bench reported as its step)."""
        rows, fs_, t0_ = engine.stream_rows(stream)
        my = list(range(nb))
        if multi and not shard_traces:
            npairs_ = nchans * (nchans - 1) // 2
            my = dist.shard_bands(dist.band_costs(npts, fs_, winlens, c['overlap'], npairs_), ngpu)[rank]
        prep = engine.prepare(nchans, npts, fs_, rij, [edges[b] for b in my], [winlens[b] for b in my], c['overlap'],
                              c['alpha'], c['ftype'], c['order'], c['ripple'])
        h.set_profiling(True)
        engine.launch(h, rows, prep, stream=engine.stream_pays(c['alpha'], prep.nwin, prep.npairs))     # (as a whole call runs it: per-batch solves, rows streamed — or one piece)
        h.sync()
        kern, stages = [], []
        for _ in range(steps):
            h.execute()
            h.sync()
            tm = h.timings()
            kern.append(tm['total_ms'])
            stages.append(tm)
        h.set_profiling(False)
        return kern, stages

    step_times = []        # wall time of every timed call of the LAST timed() run (the calls are synchronous)

    def timed(stream, steps, resident=False):
        barrier()
        t0 = time.perf_counter()
        out = None
        held = []          # the results of the timed calls stay alive until the clock has stopped: tearing down the
        del step_times[:]
        tp = t0
        for _ in range(steps):   # PREVIOUS call's 5*10^4-entry dictionary (1-2 ms) is the caller's business, not the call's
            out = one_call(stream, resident)
            tn = time.perf_counter()
            step_times.append((tn - tp) * 1e3)
            tp = tn
            if len(held) < 128:
                held.append(out)
        barrier()
        el = time.perf_counter() - t0
        del held
        if td is not None:
            import torch
            tt = torch.tensor([el], dtype=torch.float64)
            td.all_reduce(tt, op=td.ReduceOp.MAX)
            el = float(tt.item())
        return el, out

    # A band-sharded run that cannot form its communicator (or fails in the gather) is NOT replaced by something else
    # under the same metric: every rank learns of the failure, rank 0 prints a line with value null, exit code 2.
    failure = None
    try:
        for _ in range(args.warmup):
            one_call(st)
    except Exception as e:      # noqa: BLE001
        if not (multi and not shard_traces):
            raise
        failure = 'band-sharded RCCL path failed on rank %d (%s: %s)' % (rank, type(e).__name__, e)
    if td is not None:
        import torch
        flag = torch.tensor([1 if failure else 0])
        td.all_reduce(flag, op=td.ReduceOp.MAX)
        if int(flag.item()) and failure is None:
            failure = 'band-sharded RCCL path failed on another rank'
    if failure:
        print(failure, file=sys.stderr)
        give_up(failure + '; not measured (no fallback to --shard traces: that is a different, weak-scaling measurement)')
    elapsed, out = timed(st, args.steps)
    steps_ms = sorted(step_times)

    def replaced_ms(stream, steps):
        """The same calls in the loop form `out = narrow_band_least_squares(...)`: every call's result replaces the previous
        one, whose teardown (a dictionary of 5*10^4 strings, 2*10^4 small arrays) then falls INSIDE the timed loop — `value`
        keeps the results of its K calls alive until the clock has stopped (a caller in a loop pays it)."""
        barrier()
        r = one_call(stream)
        t0_ = time.perf_counter()
        for _ in range(steps):
            r = one_call(stream)
        barrier()
        dt = (time.perf_counter() - t0_) / steps * 1e3
        del r
        return dt
    ms_replaced = replaced_ms(st, max(3, args.steps // 2)) if (not multi and call_args is not None) else None

    def in_call_stage_ms(stream, reps=3):
        """Stage times INSIDE whole calls: the profiling events of every handle the call used (its band groups run on
        up to four handles of the GPU, their kernels overlap at the seams and stretch), summed over the groups —
        beside `kernel_only_ms`, which is one pass of all bands executed back to back."""
        hs = [hd for (pid, dev, slot), hd in sorted(engine._handles.items()) if pid == os.getpid()]
        for hd in hs:
            hd.set_profiling(True)
        acc = []
        try:
            for _ in range(reps):
                one_call(stream)
                tot = {}
                for hd in hs:
                    with contextlib.suppress(Exception):
                        tm = hd.timings()
                        for k in ('filter_ms', 'quantize_ms', 'screen_ms', 'verify_ms', 'solve_ms', 'xcorr_ms'):
                            tot[k] = tot.get(k, 0.0) + float(tm[k])
                acc.append(tot)
        finally:
            for hd in hs:
                hd.set_profiling(False)
        return {k: float(np.median([a_.get(k, 0.0) for a_ in acc])) for k in acc[0]} if acc else {}
    in_call = in_call_stage_ms(st) if (not multi and call_args is not None) else {}
    kern, stages = measure_kernels(st, max(3, args.steps // 2))
    nwin_list = out[6]
    units_call = int(sum(nwin_list))
    total_units = units_call * (ngpu if shard_traces else 1)
    value = total_units * args.steps / elapsed
    ms_step = elapsed / args.steps * 1e3

    line = None
    if rank == 0:
        P = nchans * (nchans - 1) // 2
        Wb = np.array([int(wl * fs) for wl in winlens], dtype=np.float64)
        incb = np.array([int(np.round((1 - c['overlap']) * w)) for w in Wb], dtype=np.float64)
        nw = np.array(nwin_list, dtype=np.float64)
        units_gpu = units_call if (not multi or shard_traces) else units_call / ngpu
        flop_total = float(np.sum(2.0 * P * Wb * Wb * nw))                    # algorithmic 2 P W^2 per unit
        bytes_total = float(np.sum((8.0 * nchans * incb + 40.0 + math.ceil(P / 8)) * nw))
        mean = lambda k: float(np.mean([s[k] for s in stages]))   # noqa: E731
        impl_used = stages[-1]['xcorr_impl']
        kernel_only = float(np.mean(kern))
        if impl_used == 3:
            kname, kern_ms, peak = 'screen_kernel (int8 MFMA screening of the full-lag correlation)', mean('screen_ms'), I8_MFMA_PEAK_TOPS
        else:
            kname = 'xcorr_mfma_kernel (f64 MFMA)' if impl_used == 2 else 'xcorr_simple_kernel'
            kern_ms, peak = mean('xcorr_ms'), FP64_MFMA_PEAK_TFLOPS
        share_f = (1.0 / ngpu) if (multi and not shard_traces) else 1.0       # this rank's part of the call
        achieved_tf = flop_total * share_f / (kern_ms * 1e-3) / 1e12
        traffic, traffic_src = None, None
        tfile = os.path.join(ROOT, 'profiles', 'traffic.json')
        if os.path.exists(tfile):
            try:
                tj = json.load(open(tfile)).get(args.config)
                if tj:
                    # per PASS (the sum over the pass's launches of the kernel), like `achieved`: VERDICT r03 item 7
                    traffic = float(tj['xcorr_hbm_bytes_per_pass']) if 'xcorr_hbm_bytes_per_pass' in tj else float(tj['xcorr_hbm_bytes_per_launch']) * int(stages[-1]['xcorr_launches'])
                    traffic_src = 'profiles/traffic.json: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command, committed (%s)' % tj.get('tag', 'r02')
            except Exception:
                traffic = None
        launches = int(stages[-1]['xcorr_launches'])
        line = {
            'metric': '(window x band) LTS solves/sec, 8-element synthetic', 'value': value, 'unit': 'solves/s',
            'n_gpus': ngpu, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': ms_step,
            'ms_per_step_min_median_max': [steps_ms[0], float(np.median(steps_ms)), steps_ms[-1]],
            'ms_per_step_results_replaced': ms_replaced,      # each call's result replaces the previous one INSIDE the timed loop (its teardown counted)
            'value_median_step': total_units / (float(np.median(steps_ms)) * 1e-3),
            'higher_is_better': True, 'scaling': 'weak' if (shard_traces or not multi) else 'strong',
            'vs_baseline': None, 'dtype': 'f64', 'data': 'synthetic',
            'config': {'workload': LABELS.get(args.config, args.config) + ', %s s windows %d%% overlap, %s order %d%s'
                                   % ('/'.join(sorted({str(w) for w in winlens}, key=float)), int(round(c['overlap'] * 100)), c['ftype'], c['order'],
                                      ' zero-phase' if c['ftype'] == 'butter' else '')
                                   + ('; band share %s (%d of %d bands)' % (share, nb, nb_all) if share else ''),
                       'step': 'one whole narrow_band_least_squares%s() call: trace upload, filter design + plan, device pass, '
                               'one D2H of grids + packed LTS weights, filter responses, stdict (%d entries)'
                               % ('_parallel' if multi and not shard_traces else '', 0 if out[4] is None else len(out[4])),
                       'units_per_call': units_call, 'bands': nb, 'elements': nchans, 'pairs': P,
                       'window_samples': sorted({int(w) for w in Wb}),
                       'parallelism': ('1 GPU' if not multi else
                                       ('%d GPUs (%s), bands of one call sharded by cost, one RCCL gather in the library'
                                        % (ngpu, 'one process drives all of them' if one_process else 'one process per GPU')
                                        if not shard_traces else '%d GPUs, one independent trace and whole call per rank, no collective' % ngpu)),
                       'scale': args.scale},
            'kernel_only_ms': kernel_only,
            'kernel_only_value': units_gpu / (kernel_only * 1e-3) * (ngpu if multi else 1),
            'stage_ms_in_call': {k.replace('_ms', ''): v for k, v in in_call.items()} or None,
            'stage_ms': {'filter': mean('filter_ms'), 'xcorr': mean('xcorr_ms'), 'solve': mean('solve_ms'),
                         'xcorr_quantize': mean('quantize_ms'), 'xcorr_screen': mean('screen_ms'),
                         'xcorr_verify': mean('verify_ms')},
            'roofline': {'bound': 'mfma', 'achieved': achieved_tf, 'peak': peak, 'unit': 'TFLOP/s',
                         'frac': achieved_tf / peak, 'traffic': traffic, 'traffic_unit': 'bytes per pass (all launches of the kernel in one pass)',
                         'traffic_source': traffic_src, 'kernel': kname,
                         'kernel_ms_per_step': kern_ms, 'launches_per_step': launches,
                         'kernel_ms_in_call': in_call.get('screen_ms') if impl_used == 3 else in_call.get('xcorr_ms'),
                         'frac_in_call': (flop_total / (in_call['screen_ms'] * 1e-3) / 1e12 / peak) if (impl_used == 3 and in_call.get('screen_ms')) else None,
                         'note': 'frac: one pass of all bands executed back to back; frac_in_call: the same kernel inside whole calls '
                                 '(sum over the call\'s band groups, HIP events per handle). ALGORITHMIC fraction: 2*P*W^2 flop per unit x units / kernel time, not issued work '
                                 '(the screening kernel issues 3 int8 limb products per multiply-add and skips lag '
                                 'blocks that cannot hold the maximum; see `noise` for the input without coherent signal)'},
            # the other compute kernel of the path, in the flops SURVEY 8(a11) prices it with (starts x 4 C-steps x 14 P):
            # what the FAST-LTS kernel delivers against the FP64 VECTOR peak (no matrix cores: 2x2 Gramians).  The honest
            # bound of this kernel is vector-instruction issue (~41 instructions per pair and selection)
            'roofline_solve': (None if c['alpha'] >= 1.0 else {
                'bound': 'fp64_vector', 'unit': 'TFLOP/s', 'peak': FP64_MFMA_PEAK_TFLOPS,
                'achieved': (min(500, P * (P - 1) // 2) * 4 * 14.0 * P) * units_gpu / (mean('solve_ms') * 1e-3) / 1e12,
                'frac': (min(500, P * (P - 1) // 2) * 4 * 14.0 * P) * units_gpu / (mean('solve_ms') * 1e-3) / 1e12 / FP64_MFMA_PEAK_TFLOPS,
                'kernel': 'solve_lts_wave_kernel (<= 8 elements)' if nchans <= 8 else 'solve_lts_bucket_kernel (9..32 elements)',
                'kernel_ms_per_step': mean('solve_ms'),
                'note': 'algorithmic flops of SURVEY 8(a11): starts * 4 * 14 * P per unit; the kernel issues several times that in selection work'}),
            'roofline_hbm': {'bound': 'hbm', 'achieved': bytes_total / (ms_step * 1e-3) / 1e9, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                             'frac': bytes_total / (ms_step * 1e-3) / 1e9 / HBM_PEAK_GBS,
                             'frac_kernel_only': bytes_total * share_f / (kernel_only * 1e-3) / 1e9 / HBM_PEAK_GBS,
                             'bytes_per_call': bytes_total,
                             'note': 'SURVEY 8(d): algorithmic bytes (8*N*inc + 40 + ceil(P/8) per unit) / wall time of the whole call'},
            'env': {k: v for k, v in sorted(os.environ.items()) if k.startswith('NBLS_')},
        }
    # what the upload of the trace costs inside a call (host-blocking copy from the stream's buffers)
    rows_up = engine.stream_rows(st)[0]
    t_up = time.perf_counter()
    for _ in range(3):
        h.set_trace_rows(rows_up, fs)
    upload_ms = (time.perf_counter() - t_up) / 3 * 1e3
    if rank == 0:
        line['upload_ms'] = upload_ms
        line['value_trace_resident'] = total_units / ((ms_step - upload_ms) * 1e-3)     # derived: the call minus the upload
    if not multi and call_args is not None:
        # ... and MEASURED: the same whole calls with the samples already in HBM when the clock starts
        # (engine.resident_trace: uploaded once, every call on the same buffers skips its upload)
        steps_r = max(3, args.steps // 2)
        with engine.resident_trace(st):
            one_call(st)
            el_r, _ = timed(st, steps_r)
        line['value_trace_resident'] = units_call * steps_r / el_r
        line['ms_per_step_trace_resident'] = el_r / steps_r * 1e3
        line['trace_resident_note'] = ('measured: whole calls inside `with resident_trace(st)` (inputs in HBM when the timed region starts); '
                                       '`value` includes the upload of the trace from the caller\'s buffers in every call')
    # ... and on incoherent noise of the same shape (no common signal: nothing for the pruning to exploit)
    if not args.no_noise and not multi:
        rng = np.random.default_rng(7)
        noise = synthetic.make_stream(rng.standard_normal((nchans, npts)), fs)
        one_call(noise)
        el_n, _ = timed(noise, max(2, args.steps // 2))
        kern_n, stages_n = measure_kernels(noise, 3)
        line['noise'] = {'value': units_call * max(2, args.steps // 2) / el_n, 'ms_per_step': el_n / max(2, args.steps // 2) * 1e3,
                         'kernel_only_ms': float(np.mean(kern_n)),
                         'xcorr_screen_ms': float(np.mean([s['screen_ms'] for s in stages_n])),
                         'xcorr_verify_ms': float(np.mean([s['verify_ms'] for s in stages_n])),
                         'input': 'independent white Gaussian noise on every element, same shape and bands'}
    # The other way to use N GPUs, beside the band-sharded call that `value` measures: every rank runs the WHOLE
    # single-GPU call on a trace of its own (another array / another stretch of time), no collective.  One call's
    # strong scaling is bounded by the host work that follows the gather (the dictionary); this is what the same
    # node delivers on independent traces.  Extra field, never `value`.
    want_ind = args.independent_calls == 'on' or (args.independent_calls == 'auto' and multi and not shard_traces and td is not None)
    if want_ind and call_args is not None and not one_process:
        ind_err, el_i = None, 0.0
        try:
            c2 = synthetic.build_config(args.config, scale=args.scale, trace_seed=synthetic.SEED + 101 + rank)
            st2 = c2['st']

            def solo():
                planner.design_cache_clear()
                with contextlib.redirect_stdout(sys.stderr):
                    return narrow_band_least_squares(*call_args[:3], st2, *call_args[4:], rij=rij)
            solo()
        except Exception as e:      # noqa: BLE001
            ind_err = '%s: %s' % (type(e).__name__, e)
        barrier()
        t_i = time.perf_counter()
        held_i = []
        if ind_err is None:
            try:
                for _ in range(args.steps):
                    held_i.append(solo())
            except Exception as e:      # noqa: BLE001
                ind_err = '%s: %s' % (type(e).__name__, e)
        barrier()
        el_i = time.perf_counter() - t_i
        del held_i
        bad_i = 1 if ind_err else 0
        if td is not None:
            import torch
            tt = torch.tensor([el_i, float(bad_i)], dtype=torch.float64)
            td.all_reduce(tt, op=td.ReduceOp.MAX)
            el_i, bad_i = float(tt[0].item()), int(tt[1].item())
        if rank == 0:
            if bad_i:
                line['independent_calls'] = {'value': None, 'note': 'failed (%s)' % (ind_err or 'on another rank')}
            else:
                line['independent_calls'] = {
                    'value': units_call * ngpu * args.steps / el_i, 'unit': 'solves/s', 'ms_per_step': el_i / args.steps * 1e3,
                    'n_gpus': ngpu, 'steps': args.steps, 'scaling': 'weak',
                    'what': 'one whole narrow_band_least_squares() call per rank on the rank\'s own synthetic trace of the same '
                            'configuration, barrier + max over ranks around the %d calls, no collective in the timed region' % args.steps}
    if rank == 0:
        if not args.no_cpu_baseline and ngpu == 1:
            line['cpu_baseline'] = cpu_baseline(c, edges, winlens)
        print(json.dumps(line), flush=True)
    if td is not None:
        td.barrier()
        td.destroy_process_group()


if __name__ == '__main__':
    main()
