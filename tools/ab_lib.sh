#!/bin/bash
# GPU box: A/B of two builds of the library on one box, alternating (DVFS / box-to-box spread make single runs incomparable).
#   tools/ab_lib.sh <tag> <libA.so> <libB.so> [rounds=2] [-- extra bench.py flags]
# A stale library (built from other sources than the tree's) is admitted for the comparison only (AFP_ALLOW_STALE_LIB).
TAG=$1; A=$2; B=$3; N=${4:-2}; shift 4 2>/dev/null; [ "$1" = "--" ] && shift
cd ${GRAFT_REPO_ROOT:-$PWD}; mkdir -p gpurun_out
Q="--steps 300 --warmup 20 --no-cpu --no-c2 --no-extras --no-host --no-table $*"
one() { AFP_LIB_PATH=$PWD/$1 AFP_ALLOW_STALE_LIB=1 python bench.py $Q 2>/dev/null | python -c "
import sys, json
b = json.loads(sys.stdin.read().strip().splitlines()[-1]); k = b['roofline']['kernels_ms']
print('%-28s %s  step %.4f ms  one-context %.4f  k_stft %.4f  k_scan %.4f  %s MHz' % (sys.argv[1], b['build_id'], b['ms_per_step'], b['ms_per_step_one_context'], k['k_stft'], k['k_scan'], b['shader_mhz_under_load']))" $1; }
for i in $(seq $N); do one $A; one $B; done | tee gpurun_out/${TAG}_ab.txt
