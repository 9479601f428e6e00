# developer: instruction counts per kernel of one cfg-3 pass (vector / scalar / LDS / matrix / vector-memory); run through
# gpurun:  gpurun -- 'bash tools/pmc_insts.sh [scale] [key=value ...]'
SCALE=${1:-1}; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/pmc_in
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES --kernel-trace --output-format csv -d gpurun_out/pmc_in -- python tools/quick_time.py cfg3 $SCALE 2 "$@" > gpurun_out/pmc_in.log 2>&1 || { tail -5 gpurun_out/pmc_in.log; exit 1; }
python - <<'PY'
import csv, glob, collections, re
f = glob.glob('gpurun_out/pmc_in/*/*counter_collection.csv')[0]
acc = collections.defaultdict(lambda: collections.defaultdict(float))
for r in csv.DictReader(open(f)):
    m = re.search(r'(\w+_kernel)', r['Kernel_Name'])
    k = m.group(1) if m else r['Kernel_Name'][:40]
    acc[k][r['Counter_Name']] += float(r['Counter_Value'])
print('instruction counts per WAVE (summed over the launches of 3 passes / waves launched)')
for k, d in sorted(acc.items(), key=lambda kv: -kv[1].get('SQ_INSTS_VALU', 0)):
    w = d.get('SQ_WAVES', 0) or 1.0
    print('%-28s waves %.3g |' % (k, w), ' '.join('%s %.0f' % (c.replace('SQ_INSTS_', ''), v / w) for c, v in sorted(d.items()) if c != 'SQ_WAVES'))
PY
