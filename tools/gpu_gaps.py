"""Developer: GPU busy/idle inside whole calls from a rocprofv3 kernel trace of tools/call_jitter.py.
usage: python tools/gpu_gaps.py <kernel_trace.csv>   (takes the last ~1/30 of the trace = the last call)"""
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted(((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']) for r in rows), key=lambda e: e[0])
# split into calls: a gap > 0.8 ms between consecutive kernel starts separates calls
calls, cur = [], [ev[0]]
for e in ev[1:]:
    if e[0] - max(x[1] for x in cur[-50:]) > 800_000:
        calls.append(cur)
        cur = []
    cur.append(e)
calls.append(cur)
print('calls found', len(calls))
for c in calls[-3:]:
    t0, t1 = c[0][0], max(x[1] for x in c)
    busy, end = 0, t0
    gaps = []
    for s, e, n in c:
        if s > end:
            gaps.append((s - end, end - t0, n))
            end_new = e
        busy += max(0, e - max(s, end))
        end = max(end, e)
    tot = sum(e - s for s, e, _ in c)
    print('span %.2f ms  union-busy %.2f ms  sum-of-kernels %.2f ms  idle %.2f ms  kernels %d' % ((t1 - t0) / 1e6, busy / 1e6, tot / 1e6, (t1 - t0 - busy) / 1e6, len(c)))
    per = {}
    for s_, e_, n_ in c:
        m_ = re.search(r'(\w+_kernel|__amd_\w+)', n_)
        k_ = m_.group(1) if m_ else n_[:28]
        per[k_] = per.get(k_, 0) + (e_ - s_)
    print('   per kernel (ms, summed over the call):', ', '.join('%s %.2f' % (k_, v_ / 1e6) for k_, v_ in sorted(per.items(), key=lambda kv: -kv[1])[:8]))
    big = sorted(gaps, reverse=True)[:3]
    for g, at, n in big:
        print('   gap %.3f ms at %.2f ms before %s' % (g / 1e6, at / 1e6, n[:50]))
