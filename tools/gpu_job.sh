#!/bin/bash
# fused ToRGB: parity, microbench, op diff per phase, and the step with/without it on the same box
R="${GRAFT_REPO_ROOT:-/root/repo}"
O=$R/gpurun_out
cd $R
timeout -k 10 300 python3 -m pytest tests/test_hip_parity.py -x -q -k "torgb" > $O/pytest_torgb.log 2>&1; echo "torgb rc=$?"
tail -5 $O/pytest_torgb.log | cut -c1-300
timeout -k 10 200 python3 tools/bench_torgb.py > $O/bench_torgb.log 2>&1; echo "bench rc=$?"
grep "bf16" $O/bench_torgb.log
timeout -k 10 400 python3 tools/op_diff_torgb.py > $O/op_diff_torgb.log 2>&1; echo "rc=$?"
grep "^##" $O/op_diff_torgb.log
for v in 1 0 1 0; do
MG_FUSED_TORGB=$v timeout -k 10 300 python3 bench.py --steps 16 --warmup 6 --no-cpu-baseline --prof-steps 0 > $O/r3_bench_b32_torgb$v.json 2> $O/r3_bench_b32_torgb$v.err; echo "bench torgb=$v rc=$?"
python3 -c "
import json;d=json.loads(open('$O/r3_bench_b32_torgb$v.json').read().strip().splitlines()[-1]);print('torgb=$v', d['ms_per_step'],d['value'])"
done
