# developer: SQ counters of the kernels of one cfg-3 pass; run through gpurun:  gpurun -- 'bash tools/pmc_screen.sh [scale]'
SCALE=${1:-0.25}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/pmc_sq
timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace --output-format csv -d gpurun_out/pmc_sq -- python tools/quick_time.py cfg3 $SCALE 2 > gpurun_out/pmc_sq.log 2>&1
python - <<'PY'
import csv, glob, collections
f = glob.glob('gpurun_out/pmc_sq/*/*counter_collection.csv')[0]
acc = collections.defaultdict(lambda: collections.defaultdict(float))
for r in csv.DictReader(open(f)):
    k = r['Kernel_Name'].split('(')[0][:40]
    acc[k][r['Counter_Name']] += float(r['Counter_Value'])
for k, d in acc.items():
    print(k, {c: '%.3g' % v for c, v in sorted(d.items())})
PY
