#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out
timeout -k 10 600 python -m pytest tests/test_hip_parity.py -x -q -m gpu > $O/r2_parity.log 2>&1
rc=$?; echo "parity rc=$rc"; tail -3 $O/r2_parity.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 400 python bench.py --no-cpu-baseline > $O/r2_b32_split.json 2> $O/r2_b32_split.err
echo "b32 rc=$?"; python -c "
import json,sys
d=json.loads(open('$O/r2_b32_split.json').read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'], d.get('roofline_mfma'))"
for b in 16 8 4; do
timeout -k 10 300 python bench.py --batch-gpu $b --no-cpu-baseline --prof-steps 0 > $O/r2_b${b}_split.json 2> $O/r2_b${b}_split.err
echo "b$b rc=$?"; python -c "
import json,sys
d=json.loads(open('$O/r2_b${b}_split.json').read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'], d['config'].get('phase_graphs'))"
done
