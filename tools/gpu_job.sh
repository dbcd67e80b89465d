#!/bin/bash
R="${GRAFT_REPO_ROOT:-/root/repo}"
cd $R
timeout -k 10 400 python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "residual_tail" > gpurun_out/wg_tests.log 2>&1; echo rc=$?; tail -12 gpurun_out/wg_tests.log | cut -c1-250
for v in 1 0 1; do
MG_FUSED_RESIDUAL=$v timeout -k 10 500 python bench.py --steps 8 --warmup 3 --no-cpu-baseline > gpurun_out/r3_bench_b32_residual$v.json 2> gpurun_out/r3_bench_b32_residual$v.err || exit 1
python - <<P
import json
d = json.loads(open('gpurun_out/r3_bench_b32_residual$v.json').read().strip().splitlines()[-1])
ak = d['roofline']['all_kernels']
print('residual=$v', d['ms_per_step'], d['value'], ak.get('residual'), ak.get('bias_act'))
P
done
