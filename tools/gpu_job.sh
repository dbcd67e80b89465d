#!/bin/bash
R="${GRAFT_REPO_ROOT:-/root/repo}"
cd $R
O=$R/gpurun_out
timeout -k 10 300 python bench.py --gpus 1 --steps 4 --warmup 2 --no-cpu-baseline --prof-steps 0 > $O/r2_bench_sanity.json 2> $O/r2_bench_sanity.err
echo "bench rc=$?"; python -c "
import json,sys
d=json.loads(open('$O/r2_bench_sanity.json').read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'], d['scaling'], d['config']['batch_per_gpu'], d['config']['phase_graphs'], d['config']['parallelism'])"
