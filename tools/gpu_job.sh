#!/bin/bash
R="${GRAFT_REPO_ROOT:-/root/repo}"
cd $R
timeout -k 10 600 python bench.py > gpurun_out/r03_bench_b32_final.json 2> gpurun_out/r03_bench_b32_final.err || exit 1
python - <<P
import json
d = json.loads(open('gpurun_out/r03_bench_b32_final.json').read().strip().splitlines()[-1])
print('default', d['ms_per_step'], d['host_enqueue_ms_per_step'], d['value'], d['roofline']['kernel'], d['roofline']['frac'], d['roofline']['achieved'])
print({k: (v['ms_per_step'], v['GBps']) for k, v in d['roofline']['all_kernels'].items() if v['ms_per_step'] > 3})
P
timeout -k 10 300 python tools/microbench.py --layout nhwc --batch 32 --dtype bfloat16 > gpurun_out/r03_microbench_bf16_nhwc_b32.log 2>&1; echo rc=$?
