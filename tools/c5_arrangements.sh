#!/bin/bash
# GPU box, VERDICT r5 #7: ONE bounded experiment on how the three C5 stages (k_stft 4.4 ms, k_scan_c 2.7, k_pairlane_ms 1.3; 8.4 ms of
# kernels, 7.7-7.9 ms pipelined) share the chip: the scan + pairing stages on a CU range of their own (--cu-split N: the
# spectral stage gets the other 256 - N compute units), against the default (all stages on all CUs), 200 steps x 2, alternating.
cd ${GRAFT_REPO_ROOT:-$PWD}; mkdir -p gpurun_out
Q="--workload c5 --steps 200 --warmup 12 --no-cpu --no-c2 --no-extras --no-host --no-table"
one() { python bench.py $Q "$@" 2>/dev/null | python -c "
import sys, json
b = json.loads(sys.stdin.read().strip().splitlines()[-1]); k = b['roofline']['kernels_ms']
print('%-34s step %.4f ms  one-context %.4f  k_stft %.3f k_scan %.3f k_pair %.3f  %s MHz' % (' '.join(sys.argv[1:]) or '(default)', b['ms_per_step'], b['ms_per_step_one_context'], k['k_stft'], k['k_scan'], k['k_pair'], b['shader_mhz_under_load']))" "$@"; }
for rep in 1 2; do
  one
  one --cu-split 32
  one --cu-split 64
  one --cu-split 96
  one --cu-split 64 --stages 2
  one --inflight 6
done | tee gpurun_out/r06_c5_arrangements.txt
