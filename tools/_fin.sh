export TMPDIR=/tmp
mkdir -p gpurun_out
bash tools/prof.sh r01g --no-overlap --no-host --no-table > gpurun_out/prof_r01g.log 2>&1
cp gpurun_out/prof_r01g/summary.txt gpurun_out/r01g_summary.txt
find gpurun_out/prof_r01g/stats -name "*kernel_stats.csv" -exec cp {} gpurun_out/r01g_kernel_stats.csv \;
rm -rf gpurun_out/prof_r01g/pmc_* gpurun_out/prof_r01g/stats
timeout 900 python bench.py > gpurun_out/bench_r01g.log 2>&1; echo "bench exit $?"
tail -1 gpurun_out/bench_r01g.log | python -c "
import sys,json; d=json.loads(sys.stdin.read())
print('c3', d['value'], d['ms_per_step'], d['ms_per_step_one_context'], d['audio_sec_per_sec'], d['roofline']['frac'], d['roofline']['whole_step_frac'], d['roofline']['kernel'], d['roofline']['kernel_ms'], d['parity'])
print(d['roofline']['kernels_ms'])
print('cpu', d['cpu_baseline']['value'], d['cpu_baseline']['audio_sec_per_sec'], d['cpu_baseline_allcores'].get('value'), d['cpu_baseline_allcores'].get('audio_sec_per_sec'))
print('c2', d['c2_single_clip']['ms'], d['c2_single_clip']['audio_sec_per_sec']); print('host', d.get('host_inclusive')); print('table', d.get('table_build'))"
for w in c4 c5; do python bench.py --workload $w --steps 10 --warmup 3 --no-cpu --no-c2 --no-host --no-table 2>&1 | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('$w', d['value'], d['ms_per_step'], d['ms_per_step_one_context'], d['audio_sec_per_sec'], d['staged'], d['batches_in_flight'])"; done
