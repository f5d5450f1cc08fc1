#!/bin/bash
# One-off evidence run (VERDICT r2 "missing" #6): the reference itself timed on the bench host's CPU, next to the oracle that
# bench.py's cpu_baseline times.  Same scratch-copy mechanism as tools/run_ref_cli_on_gpu.sh (removed afterwards).
set -e
cd /root/repo
mkdir -p _refscratch
cp /root/reference/*.py _refscratch/
trap 'rm -rf /root/repo/_refscratch' EXIT
gpurun --timeout 900 -- 'AFP_REF_DIR=/root/repo/_refscratch python tools/ref_timing.py 32 > gpurun_out/ref_timing_on_gpu_host.log 2>&1; tail -6 gpurun_out/ref_timing_on_gpu_host.log'
