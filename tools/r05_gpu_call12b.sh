mkdir -p gpurun_out; export AFP_BACKTRACE=1 PYTHONFAULTHANDLER=1 TMPDIR=/tmp
python bench.py > gpurun_out/r05_bench_builder_run.json 2> gpurun_out/s12_bench.err; echo "bench rc $?"
timeout 300 python tools/soak.py --iters 700 --reset-every 10 --tag a-shipped-final-build --log gpurun_out/r05_soak_a_final_build.log > /dev/null 2> gpurun_out/r05_soak_a2.err; echo "soak a rc $?"; tail -1 gpurun_out/r05_soak_a_final_build.log | cut -c1-200
timeout 300 python tools/soak.py --iters 700 --reset-every 10 --no-torch --tag c-system-hip-no-torch-final-build --log gpurun_out/r05_soak_c_final_build.log > /dev/null 2> gpurun_out/r05_soak_c2.err; echo "soak c rc $?"; tail -1 gpurun_out/r05_soak_c_final_build.log | cut -c1-200
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
