# final library of round 5: GPU suite, the bench line, every profile the numbers cite, a soak of the shipped configuration
mkdir -p gpurun_out; export AFP_BACKTRACE=1 PYTHONFAULTHANDLER=1 TMPDIR=/tmp
python -c "import audfprint_amd._lib as L; print('build', L.load().afp_build_id().decode())"
timeout 200 python -m pytest tests -m gpu -x -q > gpurun_out/s13_gpu_tests.log 2>&1; echo "tests rc $?"; tail -2 gpurun_out/s13_gpu_tests.log
timeout 120 python bench.py > gpurun_out/r05_bench_builder_run.json 2> gpurun_out/s13_bench.err; echo "bench rc $?"
timeout 150 bash tools/prof_all.sh r05 > gpurun_out/prof_all_r05.log 2>&1; echo "prof rc $?"; head -1 gpurun_out/prof_all_r05.log
timeout 130 python tools/soak.py --iters 700 --reset-every 10 --tag a-shipped-final-build --log gpurun_out/r05_soak_a_final_build.log > /dev/null 2> gpurun_out/r05_soak_a13.err; echo "soak a rc $?"; tail -1 gpurun_out/r05_soak_a_final_build.log | cut -c1-240
python - <<'PY'
import json
b=json.loads(open('gpurun_out/r05_bench_builder_run.json').read().strip().splitlines()[-1])
r=b['roofline']
print('c3', b['ms_per_step'], b['value'], 'roofline', {k:r.get(k) for k in ('bound','achieved','frac','traffic','kernel_ms','whole_step_frac','traffic_over_algorithmic','profile_build_id')}, b['build_id'])
print('analyzer', {k:(v['ms_per_call'], v['cut'], v['segments_rerun']) for k,v in b['analyzer_path'].items() if isinstance(v,dict)})
j=b['c4_job']; print('c4job', j['job_ms'], j['stages_ms']['download_to_host_arrays'], j['parity']['clips_checked'], j['parity']['bit_exact'], j.get('near_tie_units'))
print('table_build', {k:b['table_build'][k] for k in ('store_ms','store_kernels_ms','merge_ms','download_ms')})
print({k:(v.get('ms_per_step'), v['parity'].get('near_tie_units'), v['parity']['bit_exact']) for k,v in b.items() if isinstance(v,dict) and 'ms_per_step' in v and 'parity' in v})
PY
