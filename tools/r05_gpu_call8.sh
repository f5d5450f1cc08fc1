mkdir -p gpurun_out; export AFP_BACKTRACE=1 PYTHONFAULTHANDLER=1 TMPDIR=/tmp
python bench.py > gpurun_out/r05_bench_builder_run.json 2> gpurun_out/s8_bench.err; echo "bench rc $?"
rm -f gpurun_out/r05_soak_*.log
timeout 500 python tools/soak.py --iters 2000 --reset-every 10 --tag a-shipped --log gpurun_out/r05_soak_a_shipped.log > /dev/null 2> gpurun_out/r05_soak_a.err; echo "soak a rc $?"; tail -1 gpurun_out/r05_soak_a_shipped.log | cut -c1-260
timeout 500 python tools/soak.py --iters 2000 --reset-every 10 --no-torch --tag c-system-hip-no-torch --log gpurun_out/r05_soak_c_notorch.log > /dev/null 2> gpurun_out/r05_soak_c.err; echo "soak c rc $?"; tail -1 gpurun_out/r05_soak_c_notorch.log | cut -c1-260
AMD_SERIALIZE_KERNEL=3 AMD_SERIALIZE_COPY=3 timeout 300 python tools/soak.py --iters 500 --reset-every 10 --tag b-serialized --log gpurun_out/r05_soak_b_serialized.log > /dev/null 2> gpurun_out/r05_soak_b.err; echo "soak b rc $?"; tail -1 gpurun_out/r05_soak_b_serialized.log | cut -c1-200
GPU_MAX_HW_QUEUES=4 timeout 300 python tools/soak.py --iters 500 --reset-every 10 --tag d-4queues --log gpurun_out/r05_soak_d_4queues.log > /dev/null 2> gpurun_out/r05_soak_d.err; echo "soak d rc $?"; tail -1 gpurun_out/r05_soak_d_4queues.log | cut -c1-200
python - <<'PY'
import json
b=json.loads(open('gpurun_out/r05_bench_builder_run.json').read().strip().splitlines()[-1])
r=b['roofline']
print('c3', b['ms_per_step'], b['value'], 'roofline', {k:r.get(k) for k in ('bound','achieved','frac','traffic','kernel_ms','whole_step_frac','traffic_over_algorithmic','profile_build_id')}, r.get('valu_issue'))
print('analyzer', {k:(v['ms_per_call'], v['cut'], v['segments_rerun']) for k,v in b['analyzer_path'].items() if isinstance(v,dict)})
j=b['c4_job']; print('c4job', j['job_ms'], j['stages_ms'], j['parity']['clips_checked'], j['parity']['bit_exact'], j.get('near_tie_units'))
print({k:(v.get('ms_per_step'), v['parity'].get('near_tie_units'), v['parity']['bit_exact']) for k,v in b.items() if isinstance(v,dict) and 'ms_per_step' in v and 'parity' in v})
print('parity', b['parity'].get('bit_exact'), b['parity'].get('near_tie_units'), b['parity'].get('tie_prone_units'), 'cpu', b['cpu_baseline']['value'])
PY
