#!/bin/bash
R="${GRAFT_REPO_ROOT:-/root/repo}"
cd $R
for v in 1 0 1; do
MG_FUSED_FROMRGB=$v timeout -k 10 500 python bench.py --steps 8 --warmup 3 --no-cpu-baseline > gpurun_out/r3_bench_b32_fromrgb$v.json 2> gpurun_out/r3_bench_b32_fromrgb$v.err || exit 1
python - <<P
import json
d = json.loads(open('gpurun_out/r3_bench_b32_fromrgb$v.json').read().strip().splitlines()[-1])
ak = d['roofline']['all_kernels']
print('fromrgb=$v', d['ms_per_step'], d['value'], ak.get('fromrgb'), ak.get('bias_act'))
P
done
