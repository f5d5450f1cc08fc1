mkdir -p gpurun_out; export AFP_BACKTRACE=1 PYTHONFAULTHANDLER=1 TMPDIR=/tmp
python -m pytest tests -m gpu -x -q > gpurun_out/s4_gpu_tests.log 2>&1; echo "tests rc $?"; tail -6 gpurun_out/s4_gpu_tests.log
Q="--steps 300 --warmup 20 --no-cpu --no-c2 --no-extras --no-host --no-table"
P='import sys,json; b=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], b["ms_per_step"], b["ms_per_step_one_context"])'
for i in 1 2; do
  python bench.py $Q 2>/dev/null | python -c "$P" "c3 guard off"
  AFP_NEARTIE_EPS=1e-11 python bench.py $Q 2>/dev/null | python -c "$P" "c3 guard on "
done
# C5 k_scan backward bumps: linear table (shipped) vs de-interleaved table behind a branch tree
B="AFP_SCAN_FLAGS=-DSCAN_BWD_DEINT=1 AFP_LIB_PATH=$PWD/audfprint_amd/lib/libafp_hip_bwd_deint.so AFP_OBJ_SUFFIX=_bwd_deint"
Q5="--workload c5 --steps 200 --warmup 10 --no-cpu --no-c2 --no-extras --no-host --no-table"
for i in 1 2; do
  python bench.py $Q5 2>/dev/null | python -c "$P" "c5 linear "
  env $B python bench.py $Q5 2>/dev/null | python -c "$P" "c5 deint  "
done
Q5P="--workload c5 --steps 6 --warmup 2 --pool 128 --no-cpu --no-c2 --no-extras --no-host --no-table --no-overlap"
for v in linear deint; do
  O=$PWD/gpurun_out/prof_r05_c5_$v; mkdir -p $O
  if [ $v = deint ]; then E="env $B"; else E="env"; fi
  $E rocprofv3 --output-format csv --kernel-trace --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAVE_CYCLES -d $O/pmc -o pmc -- python bench.py $Q5P > $O/pmc.log 2>&1
  $E rocprofv3 --output-format csv --kernel-trace --stats -d $O/stats -o stats -- python bench.py $Q5P > $O/stats.log 2>&1
  python - $O $v <<'PY'
import sys, csv, glob, collections
o, v = sys.argv[1], sys.argv[2]
tot = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(int)
for f in glob.glob(o + '/pmc/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].split('(')[0][:60]
        tot[k][r['Counter_Name']] += float(r['Counter_Value'])
        n[(k, r['Counter_Name'])] += 1
for k in sorted(tot):
    if 'scan' in k:
        c = tot[k]; disp = max(1, n[(k, 'SQ_INSTS_LDS')])
        print('%s %s: dispatches %d  LDS insts/launch %.3e  bank-conflict cycles/launch %.3e  conflict cycles per LDS inst %.2f  wait_inst_lds/wave_cycles %.4f'
              % (v, k, disp, c['SQ_INSTS_LDS'] / disp, c['SQ_LDS_BANK_CONFLICT'] / disp, c['SQ_LDS_BANK_CONFLICT'] / max(1.0, c['SQ_INSTS_LDS']), c['SQ_WAIT_INST_LDS'] / max(1.0, c['SQ_WAVE_CYCLES'])))
for f in glob.glob(o + '/stats/**/*kernel_stats.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'scan' in r['Name']:
            print('%s %s avg %.1f us calls %s' % (v, r['Name'].split('(')[0][:60], float(r['AverageNs']) / 1e3, r['Calls']))
PY
  rm -rf $O
done > gpurun_out/r05_c5_bwd_table_experiment.txt 2>&1
cat gpurun_out/r05_c5_bwd_table_experiment.txt
python tools/seg_cut_sweep.py > gpurun_out/r05_seg_cut_sweep.jsonl 2> gpurun_out/r05_seg_cut_sweep.err; echo "sweep rc $?"; wc -l gpurun_out/r05_seg_cut_sweep.jsonl
