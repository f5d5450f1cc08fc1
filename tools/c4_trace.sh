#!/bin/bash
# GPU box: copy + kernel trace of bench.py's c4_job, as the library runs it now and with the round-4 changes switched off
# (uploads on each context's stream, 4 hardware queues, single-threaded download).  Usage: tools/c4_trace.sh <tag>
TAG=${1:-r04}
cd ${GRAFT_REPO_ROOT:-$PWD}
mkdir -p gpurun_out
export TMPDIR=/tmp
for mode in after before; do
  OUT=$PWD/gpurun_out/c4trace_$mode; rm -rf $OUT; mkdir -p $OUT
  if [ $mode = before ]; then export AFP_UPLOAD_STREAM=0 GPU_MAX_HW_QUEUES=4 AFP_DL_THREADS=1; else unset AFP_UPLOAD_STREAM GPU_MAX_HW_QUEUES AFP_DL_THREADS; fi
  AFP_C4_TRACE=1 timeout 400 rocprofv3 --output-format csv --kernel-trace --memory-copy-trace -d $OUT -o t -- python bench.py --steps 3 --warmup 2 --no-cpu-all --cpu-sample 4 --no-host --no-c2 --extras c4_job > $OUT/log.txt 2> $OUT/err.txt
  { echo "# $mode: AFP_UPLOAD_STREAM=${AFP_UPLOAD_STREAM:-1} GPU_MAX_HW_QUEUES=${GPU_MAX_HW_QUEUES:-12 (set by the package)} AFP_DL_THREADS=${AFP_DL_THREADS:-8}; build $(python -c 'from audfprint_amd import build; print(build.source_id())' 2>/dev/null)";
    python - $OUT/log.txt <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); c = d['c4_job']
print('# under the profiler: job_ms', c.get('job_ms'), 'parity', c.get('parity', {}).get('bit_exact'), 'stages_ms', {k: v for k, v in c.get('stages_ms', {}).items() if k != 'note'})
PY
    grep "host timeline" $OUT/err.txt | tail -1 | sed 's/^/# /';
    python tools/c4_timeline.py $OUT; } > gpurun_out/${TAG}_c4job_timeline_$mode.txt 2>&1
  rm -rf $OUT
done
head -4 gpurun_out/${TAG}_c4job_timeline_after.txt | cut -c1-250; head -4 gpurun_out/${TAG}_c4job_timeline_before.txt | cut -c1-250
