#!/bin/bash
set -o pipefail
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out
timeout -k 10 900 python -m pytest tests/test_aio_golden.py tests/test_train_step_gpu.py -q -m gpu -k "global_phase or stn or base_trainer or recorded" > $O/r2_gpu_tests2.log 2>&1
echo "pytest rc=$?"; tail -12 $O/r2_gpu_tests2.log | cut -c1-300
timeout -k 10 500 python bench.py --steps 16 --warmup 3 --no-cpu-baseline --conv-table $O/r2_conv_table_b32.csv > $O/r2_b32_default.json 2> $O/r2_b32_default.err
echo "bench rc=$?"; cut -c1-300 $O/r2_b32_default.json; head -40 $O/r2_conv_table_b32.csv
