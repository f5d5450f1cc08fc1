# the bench line of the final library with its own counter files in place (profiles/traffic.json, pmc.json of build 792293c0...) + the no-torch soak
mkdir -p gpurun_out; export AFP_BACKTRACE=1 PYTHONFAULTHANDLER=1 TMPDIR=/tmp
timeout 120 python bench.py > gpurun_out/r05_bench_builder_run.json 2> gpurun_out/s14_bench.err; echo "bench rc $?"
timeout 75 python tools/soak.py --iters 400 --reset-every 10 --no-torch --tag c-system-hip-no-torch-final-build --log gpurun_out/r05_soak_c_final_build.log > /dev/null 2> gpurun_out/r05_soak_c14.err; echo "soak c rc $?"; tail -1 gpurun_out/r05_soak_c_final_build.log | cut -c1-240
python - <<'PY'
import json
b=json.loads(open('gpurun_out/r05_bench_builder_run.json').read().strip().splitlines()[-1])
r=b['roofline']
print('c3', b['ms_per_step'], b['value'], 'roofline', {k:r.get(k) for k in ('bound','achieved','frac','traffic','kernel_ms','whole_step_frac','traffic_over_algorithmic','profile_build_id')}, b['build_id'])
PY
