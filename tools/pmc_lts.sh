# developer: SQ counters of the LTS solve kernel of one configuration; run through gpurun:
#   gpurun -- 'bash tools/pmc_lts.sh cfg5 1 [key=value ...]'      (two collection passes, 8 counters each)
CFG=${1:-cfg5}; SCALE=${2:-1}; shift; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/pmc_lts_a gpurun_out/pmc_lts_b
timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA --kernel-trace --output-format csv -d gpurun_out/pmc_lts_a -- python tools/quick_time.py $CFG $SCALE 1 "$@" > gpurun_out/pmc_lts_a.log 2>&1 || { tail -5 gpurun_out/pmc_lts_a.log; exit 1; }
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES --kernel-trace --output-format csv -d gpurun_out/pmc_lts_b -- python tools/quick_time.py $CFG $SCALE 1 "$@" > gpurun_out/pmc_lts_b.log 2>&1 || { tail -5 gpurun_out/pmc_lts_b.log; exit 1; }
python - <<'PY'
import csv, glob, collections, re
acc = collections.defaultdict(lambda: collections.defaultdict(float))
for d in ('gpurun_out/pmc_lts_a', 'gpurun_out/pmc_lts_b'):
    f = glob.glob(d + '/*/*counter_collection.csv')[0]
    for r in csv.DictReader(open(f)):
        m = re.search(r'(solve_\w+_kernel)', r['Kernel_Name'])
        if not m:
            continue
        acc[m.group(1)][r['Counter_Name']] += float(r['Counter_Value'])
for k, d in acc.items():
    wc = d.get('SQ_WAVE_CYCLES', 0) or 1.0
    print(k, {c: '%.4g' % v for c, v in sorted(d.items())})
    print('  of wave cycles: wait_any %.2f  wait_inst_any %.2f  active_any %.2f  active_valu %.2f  active_lds %.3f  active_scalar %.3f  wait_inst_lds %.3f'
          % tuple(d.get(c, 0) / wc for c in ('SQ_WAIT_ANY', 'SQ_WAIT_INST_ANY', 'SQ_ACTIVE_INST_ANY', 'SQ_ACTIVE_INST_VALU', 'SQ_ACTIVE_INST_LDS', 'SQ_ACTIVE_INST_SCA', 'SQ_WAIT_INST_LDS')))
    w = d.get('SQ_WAVES', 0) or 1.0
    print('  per wave: VALU %.0f  SALU %.0f  LDS %.0f  SMEM %.0f   lds_bank_conflict/lds_idx_active %.3f   busy_cycles %.4g'
          % (d.get('SQ_INSTS_VALU', 0) / w, d.get('SQ_INSTS_SALU', 0) / w, d.get('SQ_INSTS_LDS', 0) / w, d.get('SQ_INSTS_SMEM', 0) / w,
             d.get('SQ_LDS_BANK_CONFLICT', 0) / (d.get('SQ_LDS_IDX_ACTIVE', 0) or 1.0), d.get('SQ_BUSY_CYCLES', 0)))
PY
