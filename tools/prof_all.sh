#!/bin/bash
# GPU box: every profile the round's numbers cite, from the library in the tree.  Usage: tools/prof_all.sh <round tag, e.g. r04>
# Writes gpurun_out/<tag>_*; copy what is to be judged into profiles/ and run tools/make_traffic.py <tag>_<wl> <wl> for c3 c5 c4 c2.
TAG=${1:-r04}
cd ${GRAFT_REPO_ROOT:-$PWD}
mkdir -p gpurun_out
export TMPDIR=/tmp
bash tools/prof.sh ${TAG}_c3 --no-overlap > gpurun_out/prof_${TAG}_c3.log 2>&1
bash tools/prof.sh ${TAG}_c5 --workload c5 --no-overlap > gpurun_out/prof_${TAG}_c5.log 2>&1
bash tools/prof.sh ${TAG}_c4 --workload c4 --no-overlap > gpurun_out/prof_${TAG}_c4.log 2>&1
bash tools/prof.sh ${TAG}_c2 --workload c2 --no-overlap > gpurun_out/prof_${TAG}_c2.log 2>&1
# the headline command (4 contexts, staged): kernel trace only
OUT=$PWD/gpurun_out/prof_${TAG}_c3_overlap; mkdir -p $OUT
rocprofv3 --output-format csv --kernel-trace --stats -d $OUT/stats -o stats -- python bench.py --steps 16 --warmup 4 --pool 128 --no-cpu --no-c2 --no-extras --no-host --no-table > $OUT/stats.log 2>&1
python tools/timeline.py $OUT/stats 8 2 > gpurun_out/${TAG}_c3_overlap_timeline.txt 2>&1
find $OUT/stats -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/${TAG}_c3_overlap_kernel_stats.csv
for secs in 10 300; do
  OUT=$PWD/gpurun_out/prof_${TAG}_one$secs; mkdir -p $OUT
  rocprofv3 --output-format csv --kernel-trace --stats -d $OUT -o one -- python tools/seg_prof.py $secs > $OUT/log.txt 2>&1
  find $OUT -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/${TAG}_onefile${secs}s_kernel_stats.csv
done
rm -rf gpurun_out/prof_${TAG}_*/stats gpurun_out/prof_${TAG}_*/pmc_* gpurun_out/prof_${TAG}_one*/ 2>/dev/null
head -3 gpurun_out/${TAG}_c5_summary.txt | cut -c1-200
