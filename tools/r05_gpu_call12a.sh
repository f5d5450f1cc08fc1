mkdir -p gpurun_out; export AFP_BACKTRACE=1 PYTHONFAULTHANDLER=1 TMPDIR=/tmp
Q="--steps 5 --warmup 2 --no-cpu-all --cpu-sample 4 --no-host --no-c2 --no-extras"
P='import sys,json; b=json.loads(sys.stdin.read().strip().splitlines()[-1]); t=b["table_build"]; print(sys.argv[1], {k:t[k] for k in ("store_ms","store_kernels_ms","merge_ms","download_ms")}, t["parity"]["bit_exact"])'
python bench.py $Q 2>/dev/null | python -c "$P" "prefault 2 threads (default)"
AFP_NO_PREFAULT=1 python bench.py $Q 2>/dev/null | python -c "$P" "no prefault"
AFP_PREFAULT_THREADS=8 python bench.py $Q 2>/dev/null | python -c "$P" "prefault 8 threads"
python bench.py $Q 2>/dev/null | python -c "$P" "prefault 2 threads (default)"
python -m pytest tests -m gpu -x -q > gpurun_out/s12_gpu_tests.log 2>&1; echo "tests rc $?"; tail -3 gpurun_out/s12_gpu_tests.log
bash tools/prof_all.sh r05 > gpurun_out/prof_all_r05.log 2>&1; tail -2 gpurun_out/prof_all_r05.log | cut -c1-200
