#!/bin/bash
R="${GRAFT_REPO_ROOT:-/root/repo}"
cd $R
for b in 4 16 8; do
timeout -k 10 700 python bench.py --batch-gpu $b --no-cpu-baseline > gpurun_out/r03_bench_b${b}_final.json 2> gpurun_out/r03_bench_b${b}_final.err || exit 1
python - <<P
import json
d = json.loads(open('gpurun_out/r03_bench_b${b}_final.json').read().strip().splitlines()[-1])
pg = d['config']['phase_graphs']
print('batch $b', d['ms_per_step'], d['value'], pg if isinstance(pg, str) else (pg['recorded_and_verified'], list(pg['left_eager'].items())[:3], pg['worst_eager_spread'], pg['worst_replay_deviation']))
P
done
