#!/bin/bash
R="${GRAFT_REPO_ROOT:-/root/repo}"
cd $R
timeout -k 10 200 python tools/bench_bwd_fused.py > gpurun_out/r3_bwd_fused_bench.log 2>&1; echo rc=$?; tail -14 gpurun_out/r3_bwd_fused_bench.log
timeout -k 10 400 python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "fused or epilogue or nhwc" > gpurun_out/wg_tests.log 2>&1; echo rc=$?; tail -5 gpurun_out/wg_tests.log | cut -c1-220
