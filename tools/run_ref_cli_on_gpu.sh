#!/bin/bash
# One-off evidence run: the UNCHANGED reference CLI module (audfprint.do_cmd: precompute, new) driving the
# drop-in Analyzer on a real MI355X.  The GPU box has no /root/reference, so a scratch copy of the reference's .py files
# travels with this one gpurun call (git-ignored directory, removed again below: reference sources never enter the
# repository).  The log goes to gpurun_out/real_cli_on_gpu.log.
set -e
cd /root/repo
mkdir -p _refscratch
cp /root/reference/*.py _refscratch/
trap 'rm -rf /root/repo/_refscratch' EXIT
gpurun --timeout 600 -- 'python -m pytest tests/test_gpu_dropin.py -q -m gpu -k "cli_call_order" -s > gpurun_out/real_cli_on_gpu.log 2>&1; tail -5 gpurun_out/real_cli_on_gpu.log'
