#!/bin/bash
R="${GRAFT_REPO_ROOT:-/root/repo}"
cd $R
timeout -k 10 600 python bench.py > gpurun_out/r03_bench_b32_final.json 2> gpurun_out/r03_bench_b32_final.err || exit 1
python - <<P
import json
d = json.loads(open('gpurun_out/r03_bench_b32_final.json').read().strip().splitlines()[-1])
print('default', d['ms_per_step'], d['host_enqueue_ms_per_step'], d['value'], d['roofline']['kernel'], d['roofline']['frac'], d.get('roofline_mfma', {}).get('frac'))
P
timeout -k 10 400 python bench.py --workload single --no-cpu-baseline > gpurun_out/r03_bench_single_b32.json 2> gpurun_out/r03_bench_single_b32.err || exit 1
python - <<P
import json
d = json.loads(open('gpurun_out/r03_bench_single_b32.json').read().strip().splitlines()[-1])
print('single', d['ms_per_step'], d['value'])
P
