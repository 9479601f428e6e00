# developer: SQ counters of the kernels of one pass; run through gpurun:  gpurun -- 'bash tools/pmc_screen.sh [scale] [cfg] [key=value ...]'
SCALE=${1:-0.25}
CFG=${2:-cfg3}
shift; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/pmc_sq
timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace --output-format csv -d gpurun_out/pmc_sq -- python tools/quick_time.py $CFG $SCALE 2 "$@" > gpurun_out/pmc_sq.log 2>&1
python - <<'PY'
import csv, glob, collections, re
f = glob.glob('gpurun_out/pmc_sq/*/*counter_collection.csv')[0]
acc = collections.defaultdict(lambda: collections.defaultdict(float))
for r in csv.DictReader(open(f)):
    m = re.search(r'(\w+_kernel)', r['Kernel_Name'])
    k = m.group(1) if m else r['Kernel_Name'][:40]
    acc[k][r['Counter_Name']] += float(r['Counter_Value'])
print('SQ counters summed over the launches of 3 passes (quick_time: 1 + 2 repetitions); ratios are what matters')
for k, d in sorted(acc.items(), key=lambda kv: -kv[1].get('SQ_WAVE_CYCLES', 0)):
    wc = d.get('SQ_WAVE_CYCLES', 0) or 1.0
    print('%-28s' % k, {c: '%.3g' % v for c, v in sorted(d.items())},
          '| of wave cycles: wait_any %.2f, wait_inst_lds %.3f, lds_bank_conflict/lds_idx_active %.3f, mfma_busy/wave_cycles %.3f'
          % (d.get('SQ_WAIT_ANY', 0) / wc, d.get('SQ_WAIT_INST_LDS', 0) / wc,
             d.get('SQ_LDS_BANK_CONFLICT', 0) / (d.get('SQ_LDS_IDX_ACTIVE', 0) or 1.0), d.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / wc))
PY
