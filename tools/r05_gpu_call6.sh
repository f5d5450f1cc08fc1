mkdir -p gpurun_out; export AFP_BACKTRACE=1 PYTHONFAULTHANDLER=1 TMPDIR=/tmp
python -m pytest tests -m gpu -x -q > gpurun_out/s6_gpu_tests.log 2>&1; echo "tests rc $?"; tail -8 gpurun_out/s6_gpu_tests.log
python bench.py > gpurun_out/s6_bench.json 2> gpurun_out/s6_bench.err; echo "bench rc $?"
MASTER_ADDR=127.0.0.1 AFP_BENCH_ONE_GPU=1 AFP_BENCH_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node=2 --master-addr 127.0.0.1 --master-port 29655 bench.py --gpus 2 --steps 20 --warmup 5 > gpurun_out/r05_bench_2ranks_one_gpu.json 2> gpurun_out/s6_2ranks.err; echo "2ranks rc $?"
python - <<'PY'
import json
b=json.loads(open('gpurun_out/s6_bench.json').read().strip().splitlines()[-1])
print('c3', b['ms_per_step'], 'analyzer', {k:(v['ms_per_call'], v['cut'], v['segments_rerun']) for k,v in b['analyzer_path'].items() if isinstance(v,dict)})
j=b['c4_job']; print('c4job', j['job_ms'], j['stages_ms']['download_to_host_arrays'], j['parity']['clips_checked'], j['parity']['bit_exact'], j.get('near_tie_units'))
try:
    d=json.loads(open('gpurun_out/r05_bench_2ranks_one_gpu.json').read().strip().splitlines()[-1])
    print('2 ranks:', d['n_gpus'], d['distinct_gpus'], d['table_merge_across_ranks'])
except Exception as e: print('2ranks parse', e)
PY
