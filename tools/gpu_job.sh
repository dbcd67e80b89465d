#!/bin/bash
R="${GRAFT_REPO_ROOT:-/root/repo}"
O=$R/gpurun_out
cd $R
timeout -k 10 300 python3 -m pytest tests/test_hip_parity.py -x -q -k "modconv3x3 or fused_kernel" > $O/pytest_modconv.log 2>&1; echo "modconv rc=$?"
grep "passed\|failed\|Error" $O/pytest_modconv.log | tail -3 | cut -c1-300
timeout -k 10 200 python3 tools/bench_modconv.py > $O/bench_modconv.log 2>&1; echo "bench rc=$?"
grep -v "amdgpu\|Setting" $O/bench_modconv.log | head -6
for v in 1 0 1 0; do
MG_MODCONV=$v timeout -k 10 300 python3 bench.py --steps 16 --warmup 6 --no-cpu-baseline --prof-steps 0 > $O/r3_bench_b32_modconv$v.json 2> $O/r3_bench_b32_modconv$v.err; echo "bench modconv=$v rc=$?"
python3 -c "
import json;d=json.loads(open('$O/r3_bench_b32_modconv$v.json').read().strip().splitlines()[-1]);print('modconv=$v', d['ms_per_step'],d['value'])"
done
