#!/bin/bash
R="${GRAFT_REPO_ROOT:-/root/repo}"
cd $R
O=$R/gpurun_out
timeout -k 10 900 python -m pytest tests/test_train_step_gpu.py tests/test_hip_parity.py -x -q -m gpu -k "recorded or fused_backward or loss_phase or channels_last or bucketed" > $O/r2_parity.log 2>&1
rc=$?; echo "tests rc=$rc"; tail -3 $O/r2_parity.log; grep -n "attempt" $O/r2_parity.log | head
[ $rc -eq 0 ] || exit $rc
timeout -k 10 400 python bench.py --no-cpu-baseline --prof-steps 0 > $O/r2_b32_zp.json 2> $O/r2_b32_zp.err
echo "b32 rc=$?"; python -c "
import json,sys
d=json.loads(open('$O/r2_b32_zp.json').read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'])"
timeout -k 10 300 python bench.py --batch-gpu 4 --no-cpu-baseline --prof-steps 0 > $O/r2_b4_zp.json 2> $O/r2_b4_zp.err
echo "b4 rc=$?"; python -c "
import json,sys
d=json.loads(open('$O/r2_b4_zp.json').read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'], d['config'].get('phase_graphs',{}).get('recorded_and_verified'), list(d['config'].get('phase_graphs',{}).get('left_eager',{}).keys()))"
