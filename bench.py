#!/usr/bin/env python3
"""Headline benchmark: (window x band) LTS solves/s on the 8-element synthetic configuration
(BASELINE.json configs[2] = SURVEY.md cfg-3: 48 log bands 0.1-10 Hz, alpha 0.5, 6 h @ 40 Hz,
30 s windows / 50 % overlap, zero-phase Butterworth order 2).

    python bench.py --gpus N --steps K --warmup W

One process per GPU (N > 1: launched by torch.distributed.run, backend nccl = RCCL).  A step is
one pass of the whole hot path — filter + taper, pairwise cross-correlation / lag pick, MdCCM,
FAST-LTS + reweighting — over every (band, window) unit of the rank's bands, with the raw trace
already resident in HBM, plus (N > 1) the single all-gather of the result grids and the D2H copy
of the grids.  Scaling is weak: every rank runs the same 48 bands over its OWN six-hour trace (an
independent station-day, generator seed + rank), so per-GPU work is fixed and statistically identical,
and value = all units of all ranks / max-over-ranks time.  (Sharding the bands of ONE call over the
GPUs — strong scaling, what narrow_band_least_squares_parallel() does — leaves 6 bands = 4 ms of work per
GPU at N = 8, which measures launch latency, not the path.)

The JSON line carries `roofline` (dominant kernel = the cross-correlation; duration measured with
HIP events on the library's own stream) and `cpu_baseline` (the CPU oracle, band-parallel over
the host cores like the reference's joblib variant, on a bounded sample of the same workload).
"""
import argparse
import json
import math
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from narrow_band_least_squares_amd import dist, engine, planner, synthetic  # noqa: E402

FP64_MFMA_PEAK_TFLOPS = 78.6     # MI355X FP64 matrix = vector peak (AMD datasheet; the guide lists none)
I8_MFMA_PEAK_TOPS = 5000.0       # MI355X_MICROARCH.md: int8 MFMA = 2x the bf16 dense rate (~2.5 PF) per clock
HBM_PEAK_GBS = 8000.0            # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)


def cpu_task(args):
    """One band of the CPU oracle (the unit of the reference's joblib parallelism)."""
    sys.path.insert(0, os.path.join(ROOT, 'oracle'))
    import nbls_oracle as o
    data, fs, rij, fmin, fmax, winlen, alpha = args
    st = o.make_stream(data, fs)
    stf, _, _ = o.filter_data(st, 'butter', fmin, fmax, 2, 0.01)
    out = o.ltsva(stf, None, None, winlen, 0.5, alpha, rij=rij)
    return len(out[0])


def cpu_baseline(c, edges, budget_s=10.0):
    """Oracle ("port") timed on the host cores on a bounded sample: one band per core, the first
    `seconds` of the trace, sized from a probe so that the whole leg takes about `budget_s`."""
    from joblib import Parallel, delayed
    cores = os.cpu_count() or 1
    nb = min(cores, len(edges))
    pick = [int(round(i * (len(edges) - 1) / max(1, nb - 1))) for i in range(nb)] if nb > 1 else [len(edges) // 2]
    fs = c['fs']
    probe_s = 150.0
    t = time.time()
    n = cpu_task((c['data'][:, :int(probe_s * fs)], fs, c['rij'], edges[pick[0]][0], edges[pick[0]][1], 30.0, c['alpha']))
    per_unit = (time.time() - t) / max(1, n)
    units_per_band = max(8, int(budget_s / per_unit))
    seconds = min(c['dur'], (units_per_band + 1) * 15.0 + 30.0)
    npts = int(seconds * fs)
    tasks = [(c['data'][:, :npts], fs, c['rij'], edges[b][0], edges[b][1], 30.0, c['alpha']) for b in pick]
    t = time.time()
    counts = Parallel(n_jobs=nb)(delayed(cpu_task)(a) for a in tasks)
    wall = time.time() - t
    return {'value': sum(counts) / wall, 'unit': 'solves/s', 'cores': nb, 'kind': 'port',
            'sample': '%d bands x first %.0f s of the cfg-3 trace (%d units), NumPy oracle, joblib one band per core'
                      % (nb, seconds, sum(counts))}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=5)
    ap.add_argument('--warmup', type=int, default=1)
    ap.add_argument('--config', default='cfg3')
    ap.add_argument('--scale', type=float, default=1.0, help='shorten the trace (debug only; invalidates the metric)')
    ap.add_argument('--xcorr-impl', type=int, default=0)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    args = ap.parse_args()

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    # under torch.distributed.run the RCCL path is used even for one rank (lets the N>1 code be
    # rehearsed on a single GPU)
    use_dist = world > 1 or ('RANK' in os.environ and os.environ.get('NBLS_BENCH_FORCE_DIST', '1') == '1')
    if use_dist:
        import torch
        import torch.distributed as td
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        torch.cuda.set_device(local_rank)
        td.init_process_group('nccl', device_id=torch.device('cuda', local_rank))
    if args.gpus != world and rank == 0 and world > 1:
        print('warning: --gpus %d but WORLD_SIZE %d' % (args.gpus, world), file=sys.stderr)

    c = synthetic.build_config(args.config, scale=args.scale, trace_seed=synthetic.SEED + 1 + rank)
    bands_per_gpu = c['NBANDS']
    total_bands = bands_per_gpu
    freqlist = c['freqlist']
    all_edges = [(freqlist[i], freqlist[i + 1]) for i in range(total_bands)]
    edges = all_edges
    winlens = list(c['WINLEN_list'])

    data, fs, t0 = engine.stream_to_array(c['st'])
    nchans, npts = data.shape
    h = engine.get_handle(local_rank)
    h.set_trace(data, fs)                                  # trace resident in HBM before timing
    xij, pair_idx, xpinv = planner.co_array(c['rij'])
    h.set_geometry(xij, pair_idx, xpinv)
    P = xij.shape[0]
    W, inc, nwin = planner.window_plan(npts, fs, winlens[0], c['overlap'])
    applied = [planner.design_bandpass(c['ftype'], lo, hi, c['order'], c['ripple'], fs)[0] for lo, hi in edges]
    sos = planner.pad_sections(applied)
    tl, tr = planner.taper_ramps(npts)
    lts = planner.lts_plan(xij, c['alpha']) if c['alpha'] < 1.0 else None
    vector_len = nwin
    h.plan(sos, c['ftype'] == 'butter', tl, tr, [W] * len(edges), [inc] * len(edges), vector_len, lts=lts,
           xcorr_impl=args.xcorr_impl)
    h.set_profiling(True)
    units_rank = nwin * len(edges)

    def sync_all():
        h.sync()
        if use_dist:
            td.barrier()
            torch.cuda.synchronize()

    def step():
        h.execute()
        h.sync()
        if use_dist:
            ptrs, nbytes = h.device_results()
            dist.all_gather_device_grids(ptrs, nbytes, (len(edges), vector_len), local_rank)
        else:
            h.fetch()

    for _ in range(args.warmup):
        step()
    sync_all()
    xc, fl, sv, scr, qz, vf = [], [], [], [], [], []
    impl_used = 0
    t_start = time.perf_counter()
    for _ in range(args.steps):
        step()
        tm = h.timings()
        xc.append(tm['xcorr_ms']); fl.append(tm['filter_ms']); sv.append(tm['solve_ms'])
        scr.append(tm['screen_ms']); qz.append(tm['quantize_ms']); vf.append(tm['verify_ms'])
        impl_used = tm['xcorr_impl']
    sync_all()
    elapsed = time.perf_counter() - t_start
    if use_dist:
        tt = torch.tensor([elapsed], dtype=torch.float64, device='cuda')
        td.all_reduce(tt, op=td.ReduceOp.MAX)
        elapsed = float(tt.item())
    total_units = units_rank * world
    value = total_units * args.steps / elapsed

    if rank == 0:
        xcorr_ms = float(np.mean(xc))
        flop_unit = 2.0 * P * float(W) * float(W)
        bytes_unit = 8.0 * nchans * inc + 40.0 + math.ceil(P / 8)
        # dominant kernel: the int8-MFMA screening kernel (impl 3), else the f64-MFMA / VALU correlator
        if impl_used == 3:
            kern, kern_ms, peak = 'screen_kernel (int8 MFMA screening of the full-lag correlation)', float(np.mean(scr)), I8_MFMA_PEAK_TOPS
        else:
            kern, kern_ms, peak = 'xcorr_mfma_kernel (f64 MFMA)' if impl_used == 2 else 'xcorr_simple_kernel', xcorr_ms, FP64_MFMA_PEAK_TFLOPS
        achieved_tf = flop_unit * units_rank / (kern_ms * 1e-3) / 1e12
        achieved_gbs = bytes_unit * units_rank / (kern_ms * 1e-3) / 1e9
        traffic = None
        tf = os.path.join(ROOT, 'profiles', 'traffic.json')
        if os.path.exists(tf):
            try:
                traffic = json.load(open(tf)).get(args.config, {}).get('xcorr_hbm_bytes_per_launch')
            except Exception:
                traffic = None
        line = {
            'metric': '(window x band) LTS solves/sec, 8-element synthetic', 'value': value, 'unit': 'solves/s',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': elapsed / args.steps * 1e3, 'higher_is_better': True, 'scaling': 'weak',
            'vs_baseline': None, 'dtype': 'f64', 'data': 'synthetic',
            'config': {'workload': 'cfg-3: 8-element synthetic plane wave, %d log bands 0.1-10 Hz, LTS alpha=0.5, '
                                   '6 h @ 40 Hz, 30 s windows 50%% overlap, butter order 2 zero-phase; one '
                                   'independent 6 h trace per GPU (%d trace(s))' % (bands_per_gpu, world),
                       'units_per_gpu': units_rank, 'elements': nchans, 'pairs': P, 'window_samples': W,
                       'lts_starts': None if lts is None else int(lts['starts'].shape[0]),
                       'parallelism': '%d GPU(s), (band x window x trace) units sharded by trace, one all-gather '
                                      'of the grids' % world,
                       'scale': args.scale},
            'stage_ms': {'filter': float(np.mean(fl)), 'xcorr': xcorr_ms, 'solve': float(np.mean(sv)),
                         'xcorr_quantize': float(np.mean(qz)), 'xcorr_screen': float(np.mean(scr)),
                         'xcorr_verify': float(np.mean(vf))},
            'roofline': {'bound': 'mfma', 'achieved': achieved_tf, 'peak': peak, 'unit': 'TFLOP/s',
                         'frac': achieved_tf / peak, 'traffic': traffic, 'kernel': kern,
                         'flop_per_unit': flop_unit, 'kernel_ms_per_step': kern_ms,
                         'launches_per_step': int(h.timings()['xcorr_launches']),
                         'note': 'achieved = algorithmic 2*P*W^2 flop per unit x units / kernel time; the screening '
                                 'kernel issues 3 int8 limb products per algorithmic multiply-add (the low x low product is bounded, not computed)'},
            'roofline_hbm': {'bound': 'hbm', 'achieved': achieved_gbs, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                             'frac': achieved_gbs / HBM_PEAK_GBS, 'traffic': traffic, 'bytes_per_unit': bytes_unit},
        }
        if not args.no_cpu_baseline and world == 1:
            line['cpu_baseline'] = cpu_baseline(c, all_edges)
        print(json.dumps(line), flush=True)
    if use_dist:
        td.barrier()
        td.destroy_process_group()


if __name__ == '__main__':
    main()
