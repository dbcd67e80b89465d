#!/bin/bash
R="${GRAFT_REPO_ROOT:-/root/repo}"
cd $R
timeout -k 10 300 python tools/bench_modconv.py > gpurun_out/r3_modconv_microbench2.log 2>&1; echo rc=$?; grep -E "fused kernel|three launches|speed-up" gpurun_out/r3_modconv_microbench2.log | cut -c1-200
timeout -k 10 400 python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "modconv or wgrad or gradfix" > gpurun_out/wg_tests.log 2>&1; echo rc=$?; tail -3 gpurun_out/wg_tests.log | cut -c1-220
