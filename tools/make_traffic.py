"""Post-process the two rocprofv3 PMC passes of tools/profile_round.sh (FETCH_SIZE, WRITE_SIZE; separate runs with
--kernel-trace only, as MI355X_MICROARCH.md prescribes) into profiles/traffic.json and per-kernel summaries.
Counter values are KiB; FETCH_SIZE is doubled (gfx950 reports half of a wide 16-B/lane coalesced stream).
Usage: make_traffic.py TAG DIR --traffic-only"""
import collections
import csv
import glob
import json
import os
import re
import sys


def kname(full):
    m = re.search(r'(\w+_kernel)', full)
    return m.group(1) if m else full[:40]


def pmc(d, counter):
    f = glob.glob(d + '/*/*counter_collection.csv')[0]
    tot = collections.defaultdict(float)
    n = collections.Counter()
    for r in csv.DictReader(open(f)):
        if r['Counter_Name'] != counter:
            continue
        k = kname(r['Kernel_Name'])
        tot[k] += float(r['Counter_Value']) * 1024.0
        n[k] += 1
    return tot, n


def main():
    tag, d = sys.argv[1], sys.argv[2]
    fetch, nf = pmc(d + '/fetch', 'FETCH_SIZE')
    write, nw = pmc(d + '/write', 'WRITE_SIZE')
    # the bench command of profile_round.sh: warmup 1 + steps 4 whole calls (3 band-group passes each) + 1 planning
    # pass + 3 kernel-only passes of all bands = 9 passes over all units; screen_kernel launches tell the exact count
    dom = 'screen_kernel'
    per_kernel = {}
    for k in sorted(set(fetch) | set(write)):
        per_kernel[k] = {'FETCH_SIZE_bytes_total': fetch.get(k, 0.0), 'WRITE_SIZE_bytes_total': write.get(k, 0.0),
                         'launches': int(nf.get(k, nw.get(k, 0)))}
    # passes over all units = total units screened / units per pass: every pass quantises each (unit, channel) once,
    # so use the solve kernels' launch count (one per pass) when present
    # one forward state kernel per device pass (whole calls are ONE streamed pass since round 4)
    passes = max(1, int(nf.get('filter_state_mfma_kernel', 0) or nf.get('filter_state_kernel', 0) or 1))
    js = {'tag': tag, 'kernel': dom, 'passes': passes,
          'xcorr_hbm_bytes_per_pass': (2.0 * fetch.get(dom, 0.0) + write.get(dom, 0.0)) / passes,
          'all_kernels_hbm_bytes_per_pass': sum(2.0 * fetch.get(k, 0.0) + write.get(k, 0.0) for k in per_kernel) / passes,
          'xcorr_hbm_bytes_per_launch': (2.0 * fetch.get(dom, 0.0) + write.get(dom, 0.0)) / max(1, nf.get(dom, 1)),
          'all_kernels_hbm_bytes_total': sum(2.0 * fetch.get(k, 0.0) + write.get(k, 0.0) for k in per_kernel),
          'method': 'rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes of `python bench.py --steps 4 --warmup 1 '
                    '--no-cpu-baseline --no-noise` (with --kernel-trace only); counter values are KiB; FETCH_SIZE doubled per '
                    'MI355X_MICROARCH.md (gfx950 reports half of a wide 16-B/lane coalesced stream); per PASS = the sum over all '
                    'launches of the kernel / the number of device passes in the run (one forward filter-state kernel per pass); '
                    'tools/pmc_traffic.sh gives the per-pass sums of every kernel',
          'per_kernel': per_kernel}
    root = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
    old = {}
    path = os.path.join(root, 'profiles', 'traffic.json')
    if os.path.exists(path):
        try:
            old = json.load(open(path))
        except Exception:
            old = {}
    old = {k: v for k, v in old.items() if k.startswith('cfg')}
    old['cfg3'] = js
    json.dump(old, open(path, 'w'), indent=1)
    json.dump(old, open(d + '/traffic.json', 'w'), indent=1)
    print('traffic per pass (%s): %.1f MB over %d passes' % (dom, js['xcorr_hbm_bytes_per_pass'] / 1e6, passes))


if __name__ == '__main__':
    main()
