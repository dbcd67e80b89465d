#!/bin/bash
R="${GRAFT_REPO_ROOT:-/root/repo}"
cd $R
for v in 1 0 1 0; do
MG_FINISH_KERNEL=$v timeout -k 10 500 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --prof-steps 0 > gpurun_out/r3_bench_b32_finish$v.json 2> gpurun_out/r3_e.err || exit 1
python - <<P
import json
d = json.loads(open('gpurun_out/r3_bench_b32_finish$v.json').read().strip().splitlines()[-1])
print('finish=$v', d['ms_per_step'], d['host_enqueue_ms_per_step'], d['value'])
P
done
