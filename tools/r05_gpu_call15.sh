# final library: the unchanged reference CLI on the GPU (scratch copy of the reference's .py files for this call only), smoke(), and the
# driver's own bench invocation
mkdir -p gpurun_out; export AFP_BACKTRACE=1 PYTHONFAULTHANDLER=1 TMPDIR=/tmp
python -c "import audfprint_amd._lib as L; print('build', L.load().afp_build_id().decode())" > gpurun_out/r05_real_cli_on_gpu.log
timeout 60 python -m pytest tests/test_gpu_dropin.py -q -m gpu -k "cli_call_order" -s >> gpurun_out/r05_real_cli_on_gpu.log 2>&1; echo "cli rc $?"; tail -2 gpurun_out/r05_real_cli_on_gpu.log
timeout 60 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
S=$(date +%s.%N); timeout 100 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r05_bench_driver_style.json 2> gpurun_out/s15_bench.err; echo "bench rc $? wall $(echo "$(date +%s.%N) - $S" | bc) s"
python - <<'PY'
import json
b=json.loads(open('gpurun_out/r05_bench_driver_style.json').read().strip().splitlines()[-1])
print(b['ms_per_step'], b['value'], b['roofline']['frac'], b['roofline']['traffic'], b['c4_job']['job_ms'], b['c4_job']['parity']['clips_checked'], b['c4_job']['parity']['bit_exact'])
PY
