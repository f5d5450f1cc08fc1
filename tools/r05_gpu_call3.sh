mkdir -p gpurun_out; export AFP_BACKTRACE=1 PYTHONFAULTHANDLER=1
python -m pytest tests -m gpu -x -q > gpurun_out/s3_gpu_tests.log 2>&1; echo "tests rc $?"; tail -12 gpurun_out/s3_gpu_tests.log
Q="--steps 300 --warmup 20 --no-cpu --no-c2 --no-extras --no-host --no-table"
for i in 1 2; do
  AFP_NEARTIE_EPS=0 python bench.py $Q 2>/dev/null | python -c "import sys,json; b=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('guard off', b['ms_per_step'], b['ms_per_step_one_context'])"
  python bench.py $Q 2>/dev/null | python -c "import sys,json; b=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('guard on ', b['ms_per_step'], b['ms_per_step_one_context'])"
done
AFP_COMPACT=0 AFP_NEARTIE_EPS=0 python bench.py $Q 2>/dev/null | python -c "import sys,json; b=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('dense guard off', b['ms_per_step'], b['ms_per_step_one_context'])"
AFP_COMPACT=0 python bench.py $Q 2>/dev/null | python -c "import sys,json; b=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('dense guard on ', b['ms_per_step'], b['ms_per_step_one_context'])"
python bench.py > gpurun_out/s3_bench.json 2> gpurun_out/s3_bench.err; echo "bench rc $?"
python tools/analyzer_breakdown.py 10 30 > gpurun_out/s3_breakdown.txt 2>&1; tail -30 gpurun_out/s3_breakdown.txt
timeout 400 python tools/soak.py --iters 2000 --reset-every 10 --tag a-shipped --log gpurun_out/r05_soak_a_shipped.log > /dev/null 2> gpurun_out/r05_soak_a.err; echo "soak a rc $?"; tail -2 gpurun_out/r05_soak_a_shipped.log | cut -c1-300
timeout 400 python tools/soak.py --iters 2000 --reset-every 10 --no-torch --tag c-system-hip-no-torch --log gpurun_out/r05_soak_c_notorch.log > /dev/null 2> gpurun_out/r05_soak_c.err; echo "soak c rc $?"; tail -2 gpurun_out/r05_soak_c_notorch.log | cut -c1-300
AMD_SERIALIZE_KERNEL=3 AMD_SERIALIZE_COPY=3 timeout 300 python tools/soak.py --iters 500 --reset-every 10 --tag b-serialized --log gpurun_out/r05_soak_b_serialized.log > /dev/null 2> gpurun_out/r05_soak_b.err; echo "soak b rc $?"; tail -1 gpurun_out/r05_soak_b_serialized.log | cut -c1-300
GPU_MAX_HW_QUEUES=4 timeout 300 python tools/soak.py --iters 500 --reset-every 10 --tag d-4queues --log gpurun_out/r05_soak_d_4queues.log > /dev/null 2> gpurun_out/r05_soak_d.err; echo "soak d rc $?"; tail -1 gpurun_out/r05_soak_d_4queues.log | cut -c1-300
