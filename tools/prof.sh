#!/bin/bash
# GPU box: kernel-trace stats + PMC passes for one bench workload.  Usage: tools/prof.sh <tag> [bench args]
#   e.g. tools/prof.sh r02_c3 --no-overlap            (kernels strictly one after another: per-kernel figures)
#        tools/prof.sh r02_c3_overlap                 (4 contexts in flight: what the headline number runs)
# rocprofv3 passes are separate runs (counters never share a run with --stats; FETCH_SIZE and WRITE_SIZE do not fit one pass).
set -u
TAG=${1:-r02}; shift || true
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p $OUT
ARGS="--steps 6 --warmup 2 --pool 128 --no-cpu --no-c2 --no-extras --no-host --no-table $*"
rocprofv3 --output-format csv --kernel-trace --stats -d $OUT/stats -o stats -- python bench.py $ARGS > $OUT/stats.log 2>&1
rocprofv3 --output-format csv --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT -d $OUT/pmc_sq -o pmc -- python bench.py $ARGS > $OUT/pmc_sq.log 2>&1
rocprofv3 --output-format csv --kernel-trace --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_LDS_IDX_ACTIVE -d $OUT/pmc_sq2 -o pmc -- python bench.py $ARGS > $OUT/pmc_sq2.log 2>&1
rocprofv3 --output-format csv --kernel-trace --pmc FETCH_SIZE -d $OUT/pmc_fetch -o pmc -- python bench.py $ARGS > $OUT/pmc_fetch.log 2>&1
rocprofv3 --output-format csv --kernel-trace --pmc WRITE_SIZE -d $OUT/pmc_write -o pmc -- python bench.py $ARGS > $OUT/pmc_write.log 2>&1
python tools/prof_summary.py $OUT > $OUT/summary.txt 2>&1
python tools/timeline.py $OUT/stats 8 2 > $OUT/timeline.txt 2>&1
python tools/timeline.py $OUT/pmc_fetch 8 2 > $OUT/timeline_under_pmc.txt 2>&1
cp $OUT/summary.txt $PWD/gpurun_out/${TAG}_summary.txt
cp $OUT/timeline.txt $PWD/gpurun_out/${TAG}_timeline.txt
cp $OUT/timeline_under_pmc.txt $PWD/gpurun_out/${TAG}_timeline_under_pmc.txt
find $OUT/stats -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $PWD/gpurun_out/${TAG}_kernel_stats.csv
tail -40 $OUT/summary.txt
