#!/bin/bash
R="${GRAFT_REPO_ROOT:-/root/repo}"
cd $R
timeout -k 10 600 python bench.py > gpurun_out/r03_bench_b32_final.json 2> gpurun_out/r03_bench_b32_final.err || exit 1
python - <<P
import json
d = json.loads(open('gpurun_out/r03_bench_b32_final.json').read().strip().splitlines()[-1])
print('default', d['ms_per_step'], d['host_enqueue_ms_per_step'], d['value'], d['roofline']['kernel'], d['roofline']['frac'], d.get('roofline_mfma', {}).get('frac'))
P
grep -E "step (8|12|16|20|24|28|32)/32" gpurun_out/r03_bench_b32_final.err | cut -c1-80
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > gpurun_out/r3_gpu_tests.log 2>&1; echo rc=$?; tail -3 gpurun_out/r3_gpu_tests.log | cut -c1-220
