#!/bin/bash
# GPU box: kernel-trace stats + PMC passes for the bench workload.  Usage: tools/prof.sh <tag> [bench args]
set -u
TAG=${1:-r01}; shift || true
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p $OUT
ARGS="--steps 5 --warmup 2 --no-cpu --no-c2 $*"
rocprofv3 --output-format csv --kernel-trace --stats -d $OUT/stats -o stats -- python bench.py $ARGS > $OUT/stats.log 2>&1
rocprofv3 --output-format csv --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT -d $OUT/pmc_sq -o pmc -- python bench.py $ARGS > $OUT/pmc_sq.log 2>&1
rocprofv3 --output-format csv --kernel-trace --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_LDS_IDX_ACTIVE -d $OUT/pmc_sq2 -o pmc -- python bench.py $ARGS > $OUT/pmc_sq2.log 2>&1
rocprofv3 --output-format csv --kernel-trace --pmc FETCH_SIZE -d $OUT/pmc_fetch -o pmc -- python bench.py $ARGS > $OUT/pmc_fetch.log 2>&1
rocprofv3 --output-format csv --kernel-trace --pmc WRITE_SIZE -d $OUT/pmc_write -o pmc -- python bench.py $ARGS > $OUT/pmc_write.log 2>&1
find $OUT -name "*.csv" | head -30
python tools/prof_summary.py $OUT > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
