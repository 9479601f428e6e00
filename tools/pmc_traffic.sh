# developer: HBM-side bytes per kernel of one pass (FETCH_SIZE and WRITE_SIZE in separate rocprofv3 passes, as
# MI355X_MICROARCH.md prescribes; FETCH_SIZE doubled: gfx950 reports half of a wide coalesced stream).
#   gpurun -- 'bash tools/pmc_traffic.sh cfg3 1'
CFG=${1:-cfg3}; SCALE=${2:-1}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/pt_f gpurun_out/pt_w
timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pt_f -- python tools/quick_time.py $CFG $SCALE 2 > gpurun_out/pt_f.log 2>&1 || exit 3
timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pt_w -- python tools/quick_time.py $CFG $SCALE 2 > gpurun_out/pt_w.log 2>&1 || exit 4
python - <<'PY'
import csv, glob, collections, re
def load(d, name):
    f = glob.glob(d + '/*/*counter_collection.csv')[0]
    tot = collections.defaultdict(float); n = collections.Counter()
    for r in csv.DictReader(open(f)):
        if r['Counter_Name'] != name: continue
        m = re.search(r'(\w+_kernel)', r['Kernel_Name']); k = m.group(1) if m else r['Kernel_Name'][:30]
        tot[k] += float(r['Counter_Value']) * 1024.0; n[k] += 1
    return tot, n
f, nf = load('gpurun_out/pt_f', 'FETCH_SIZE'); w, nw = load('gpurun_out/pt_w', 'WRITE_SIZE')
passes = 3.0        # quick_time: one launch + 2 repetitions of the same pass
tot = 0.0
for k in sorted(set(f) | set(w), key=lambda k: -(2 * f.get(k, 0) + w.get(k, 0))):
    b = (2 * f.get(k, 0) + w.get(k, 0)) / passes
    tot += b
    print('%-32s launches/pass %5.1f  fetch(x2) %8.3f GB  write %8.3f GB  total %8.3f GB per pass' % (k, nf.get(k, 0) / passes, 2 * f.get(k, 0) / passes / 1e9, w.get(k, 0) / passes / 1e9, b / 1e9))
print('ALL KERNELS: %.2f GB per pass' % (tot / 1e9))
PY
