mkdir -p gpurun_out; export AFP_BACKTRACE=1 PYTHONFAULTHANDLER=1 TMPDIR=/tmp
for mode in after dense_download; do
  OUT=$PWD/gpurun_out/c4trace_$mode; rm -rf $OUT; mkdir -p $OUT
  if [ $mode = dense_download ]; then export AFP_TABLE_DENSE_DOWNLOAD=1 AFP_NO_PREFAULT=1; else unset AFP_TABLE_DENSE_DOWNLOAD AFP_NO_PREFAULT; fi
  AFP_C4_TRACE=1 timeout 400 rocprofv3 --output-format csv --kernel-trace --memory-copy-trace -d $OUT -o t -- python bench.py --steps 3 --warmup 2 --no-cpu-all --cpu-sample 4 --no-host --no-c2 --extras c4_job > $OUT/log.txt 2> $OUT/err.txt
  { echo "# $mode: AFP_TABLE_DENSE_DOWNLOAD=${AFP_TABLE_DENSE_DOWNLOAD:-0} AFP_NO_PREFAULT=${AFP_NO_PREFAULT:-0}; build $(python -c 'from audfprint_amd import build; print(build.source_id())' 2>/dev/null)";
    python - $OUT/log.txt <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); c = d['c4_job']
print('# under the profiler: job_ms', c.get('job_ms'), 'parity', c.get('parity', {}).get('bit_exact'), 'clips', c.get('parity', {}).get('clips_checked'), 'bytes downloaded', c.get('table_bytes_downloaded'), 'stages_ms', {k: v for k, v in c.get('stages_ms', {}).items() if k != 'note'})
PY
    grep "host timeline" $OUT/err.txt | tail -1 | sed 's/^/# /';
    python tools/c4_timeline.py $OUT; } > gpurun_out/r05_c4job_timeline_$mode.txt 2>&1
  rm -rf $OUT
done
unset AFP_TABLE_DENSE_DOWNLOAD AFP_NO_PREFAULT
head -3 gpurun_out/r05_c4job_timeline_after.txt | cut -c1-400; head -3 gpurun_out/r05_c4job_timeline_dense_download.txt | cut -c1-400
timeout 1300 python tools/soak.py --iters 8000 --reset-every 10 --tag a-shipped-long --log gpurun_out/r05_soak_a_shipped_8000.log > /dev/null 2> gpurun_out/r05_soak_a8000.err; echo "soak long rc $?"; tail -1 gpurun_out/r05_soak_a_shipped_8000.log | cut -c1-200
