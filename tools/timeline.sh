#!/bin/bash
# GPU box: kernel trace of the overlapped bench -> tools/timeline.py summary.  Usage: tools/timeline.sh <tag> [bench args]
TAG=${1:-tl}; shift || true
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/tl_$TAG
mkdir -p $OUT
rocprofv3 --output-format csv --kernel-trace -d $OUT -o tl -- python bench.py --steps 24 --warmup 4 --no-cpu --no-extras --no-c2 --no-host --no-table $* > $OUT/run.log 2>&1
python tools/timeline.py $OUT 8 4 > $OUT/timeline.txt 2>&1
cat $OUT/timeline.txt
