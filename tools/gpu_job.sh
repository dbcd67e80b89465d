#!/bin/bash
R="${GRAFT_REPO_ROOT:-/root/repo}"
cd $R
O=$R/gpurun_out
timeout -k 10 900 python -m pytest tests/test_hip_parity.py tests/test_train_step_gpu.py tests/test_aio_golden.py -x -q -m gpu -k "minibatch or golden or loss_phase or forward_matches or global_phase" > $O/r2_parity.log 2>&1
rc=$?; echo "tests rc=$rc"; tail -3 $O/r2_parity.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 400 python bench.py --no-cpu-baseline > $O/r2_b32_mb.json 2> $O/r2_b32_mb.err
echo "b32 rc=$?"; python -c "
import json,sys
d=json.loads(open('$O/r2_b32_mb.json').read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'], d['roofline']['all_kernels'].get('mbstd'))"
