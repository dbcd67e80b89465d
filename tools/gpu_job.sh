#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out
timeout -k 10 600 python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "warp or augment" > $O/r2_parity.log 2>&1
rc=$?; echo "parity rc=$rc"; tail -3 $O/r2_parity.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 400 python bench.py --no-cpu-baseline --steps 8 --warmup 4 > $O/r2_b32_warp.json 2> $O/r2_b32_warp.err
echo "b32 rc=$?"; python -c "
import json,sys
d=json.loads(open('$O/r2_b32_warp.json').read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value']); print({k:(v['ms_per_step'],v['launches_per_step'],v['GBps']) for k,v in d['roofline']['all_kernels'].items() if 'warp' in k or 'sep' in k})"
