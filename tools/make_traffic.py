"""Post-process the rocprofv3 output of tools/profile_round.sh into the summaries kept under profiles/
(kernel stats CSV, per-kernel PMC means, traffic.json).  Usage: make_traffic.py TAG DIR"""
import collections
import csv
import glob
import json
import re
import shutil
import sys


def kname(full):
    m = re.search(r'(\w+_kernel)', full)
    return m.group(1) if m else full[:40]


def pmc_means(d, counter):
    f = glob.glob(d + '/*/*counter_collection.csv')[0]
    tot = collections.defaultdict(float)
    n = collections.Counter()
    for r in csv.DictReader(open(f)):
        if r['Counter_Name'] != counter:
            continue
        k = kname(r['Kernel_Name'])
        tot[k] += float(r['Counter_Value'])
        n[k] += 1
    return {k: (tot[k] / n[k], n[k]) for k in tot}


def main():
    tag, d = sys.argv[1], sys.argv[2]
    traffic_only = len(sys.argv) > 3 and sys.argv[3] == '--traffic-only'
    out = d + '/out'
    import os
    os.makedirs(out, exist_ok=True)
    if not traffic_only:
        shutil.copy(glob.glob(d + '/stats/*/*kernel_stats.csv')[0], '%s/%s_bench_cfg3_kernel_stats.csv' % (out, tag))
        shutil.copy('%s/%s_bench_cfg3.json' % (d, tag), '%s/%s_bench_cfg3.json' % (out, tag))
        print(open('%s/%s_bench_cfg3.json' % (out, tag)).read().strip()[:600])
        return
    fetch = pmc_means(d + '/fetch', 'FETCH_SIZE')
    write = pmc_means(d + '/write', 'WRITE_SIZE')
    for name, tab in (('fetch', fetch), ('write', write)):
        with open('%s/%s_pmc_%s_size_summary.csv' % (out, tag, name), 'w') as fh:
            fh.write('kernel,launches,mean_KiB_per_launch\n')
            for k, (v, c) in sorted(tab.items()):
                fh.write('%s,%d,%.3f\n' % (k, c, v))
    raw = {}
    for k in fetch:
        raw[k] = {'FETCH_SIZE_bytes': fetch[k][0] * 1024.0, 'WRITE_SIZE_bytes': write.get(k, (0.0, 0))[0] * 1024.0,
                  'launches': fetch[k][1]}
    dom = 'screen_kernel'
    traffic = 2.0 * raw[dom]['FETCH_SIZE_bytes'] + raw[dom]['WRITE_SIZE_bytes']
    js = {'cfg3': {'xcorr_hbm_bytes_per_launch': traffic, 'kernel': dom,
                   'method': 'rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes (bench.py --steps 2 '
                             '--warmup 1, with --kernel-trace only); counter values are KiB; FETCH_SIZE doubled per '
                             'MI355X_MICROARCH.md (gfx950 reports half of a wide 16-B/lane coalesced stream; the staging '
                             'loads of this kernel are 16 B/lane); per-launch mean over the unit batches of a step',
                   'raw_per_launch_bytes': raw}}
    json.dump(js, open(out + '/traffic.json', 'w'), indent=1)
    # bench.py reads profiles/traffic.json: refresh it BEFORE the bench run of this script
    json.dump(js, open(os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'profiles', 'traffic.json'), 'w'), indent=1)
    print('traffic per launch (%s): %.1f MB' % (dom, traffic / 1e6))


if __name__ == '__main__':
    main()
