#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out
timeout -k 10 300 python -m pytest tests/test_train_step_gpu.py -x -q -m gpu -k "bucketed or base_trainer" > $O/r2_gpu_tests2.log 2>&1
rc=$?; echo "gpu tests rc=$rc"; tail -5 $O/r2_gpu_tests2.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 400 python bench.py --no-cpu-baseline > $O/r2_b32_padup.json 2> $O/r2_b32_padup.err
echo "b32 rc=$?"; python -c "
import json,sys
d=json.loads(open('$O/r2_b32_padup.json').read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value']); print({k:(v['ms_per_step'],v['launches_per_step']) for k,v in d['roofline']['all_kernels'].items()})"
