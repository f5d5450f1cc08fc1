#!/bin/bash
# The one script a gpurun call runs on the GPU box (replaces the per-call scratch scripts of earlier rounds):
#     gpurun --timeout 1500 -- 'bash tools/gpu_call.sh r06a tests bench prof'
# usage: tools/gpu_call.sh <tag> <step> [<step> ...]      everything lands in gpurun_out/<tag>_*
#   tests        python -m pytest tests -m gpu                      -> <tag>_gpu_tests.log
#   bench        python bench.py (the default, extras included)     -> <tag>_bench.json + a one-screen summary
#   driver       python bench.py --gpus 1 --steps 20 --warmup 5     -> <tag>_bench_driver_style.json (what the driver runs)
#   prof         tools/prof_all.sh <tag> (kernel trace + PMC passes of c3 / c5 / c4 / c2, one-file traces)
#   soak[:N]     tools/soak.py, N iterations (default 700) of the shipped configuration
#   kres         kernel resource usage of k_stft.hip / k_scan.hip (needs no GPU, kept here for one-stop logs)
#   smoke        __graft_entry__.smoke()
#   py:<file>    python <file> (a tools/ script), output -> <tag>_<basename>.log
TAG=$1; shift
cd ${GRAFT_REPO_ROOT:-$PWD}
mkdir -p gpurun_out; export AFP_BACKTRACE=1 PYTHONFAULTHANDLER=1 TMPDIR=/tmp
python -c "import audfprint_amd._lib as L; print('build', L.load().afp_build_id().decode())"
summary() {
python - "$1" <<'PY'
import json, sys
b = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r = b['roofline']
print('c3', b['ms_per_step'], 'ms', b['value'], 'hashes/s | roofline', {k: r.get(k) for k in ('bound', 'achieved', 'frac', 'traffic', 'kernel_ms', 'whole_step_frac', 'traffic_over_algorithmic', 'profile_build_id')}, b['build_id'])
p = b.get('parity', {})
print('parity', {k: p.get(k) for k in ('clips_checked', 'bit_exact', 'timed_variant_checked', 'guarded_pass_identical', 'near_tie_units', 'tie_prone_units')})
print('cpu', {k: b.get(k, {}).get('value') for k in ('cpu_baseline', 'cpu_baseline_allcores')}, b.get('cpu_baseline', {}).get('kind'), b.get('cpu_baseline_allcores', {}).get('cores'))
if 'analyzer_path' in b:
    print('analyzer', {k: (v['ms_per_call'], v['bit_exact'], v.get('segments_rerun')) for k, v in b['analyzer_path'].items() if isinstance(v, dict)})
if 'c4_job' in b and 'job_ms' in b['c4_job']:
    j = b['c4_job']
    print('c4job', j['job_ms'], j['stages_ms'], j['parity']['clips_checked'], j['parity']['bit_exact'])
if 'table_build' in b:
    print('table_build', {k: b['table_build'][k] for k in ('store_ms', 'store_kernels_ms', 'merge_ms', 'download_ms')}, b['table_build'].get('parity', {}).get('bit_exact'))
for k, v in b.items():
    if isinstance(v, dict) and 'parity' in v and k != 'table_build':
        pp = v['parity']
        print(k, v.get('ms_per_step', v.get('ms', v.get('job_ms', v.get('ms_per_query')))), {q: pp.get(q) for q in ('bit_exact', 'timed_variant_checked', 'guarded_pass_identical', 'near_tie_units', 'clips_checked')})
for k in b:
    if k.endswith('error'):
        print('ERROR', k, b[k])
PY
}
for step in "$@"; do
  case $step in
    tests)  timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/${TAG}_gpu_tests.log 2>&1; echo "tests rc $?"; tail -4 gpurun_out/${TAG}_gpu_tests.log ;;
    bench)  timeout 300 python bench.py > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err; echo "bench rc $?"; tail -3 gpurun_out/${TAG}_bench.err; summary gpurun_out/${TAG}_bench.json ;;
    driver) T0=$SECONDS; timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/${TAG}_bench_driver_style.json 2> gpurun_out/${TAG}_bench_driver_style.err; echo "driver rc $? (whole command $((SECONDS - T0)) s)"; tail -2 gpurun_out/${TAG}_bench_driver_style.err; summary gpurun_out/${TAG}_bench_driver_style.json ;;
    prof)   timeout 400 bash tools/prof_all.sh ${TAG} > gpurun_out/prof_all_${TAG}.log 2>&1; echo "prof rc $?"; head -3 gpurun_out/prof_all_${TAG}.log ;;
    soak*)  N=${step#soak:}; [ "$N" = "soak" ] && N=700
            timeout 400 python tools/soak.py --iters $N --reset-every 10 --tag ${TAG}-shipped --log gpurun_out/${TAG}_soak_shipped.log > /dev/null 2> gpurun_out/${TAG}_soak.err; echo "soak rc $?"; tail -1 gpurun_out/${TAG}_soak_shipped.log | cut -c1-240 ;;
    kres)   for f in k_stft.hip k_scan.hip; do bash tools/kres.sh $f; done > gpurun_out/${TAG}_kres.txt 2>&1; echo "kres rc $?" ;;
    smoke)  timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/${TAG}_smoke.log 2>&1; echo "smoke rc $?"; tail -1 gpurun_out/${TAG}_smoke.log ;;
    py:*)   f=${step#py:}; timeout 600 python $f > gpurun_out/${TAG}_$(basename ${f%.py}).log 2>&1; echo "$f rc $?"; tail -5 gpurun_out/${TAG}_$(basename ${f%.py}).log ;;
    *)      echo "unknown step $step" ;;
  esac
done
