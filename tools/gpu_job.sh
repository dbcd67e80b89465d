#!/bin/bash
R="${GRAFT_REPO_ROOT:-/root/repo}"
cd $R
for v in hipblaslt rocblas hipblaslt; do
MG_BLAS=$v timeout -k 10 500 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --prof-steps 0 > gpurun_out/r3_bench_b32_blas_$v.json 2> gpurun_out/r3_bench_b32_blas_$v.err || exit 1
python - <<P
import json
d = json.loads(open('gpurun_out/r3_bench_b32_blas_$v.json').read().strip().splitlines()[-1])
print('MG_BLAS=$v', d['ms_per_step'], d['host_enqueue_ms_per_step'], d['value'], d['config']['blas'][:30])
P
done
