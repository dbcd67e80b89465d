#!/bin/bash
R="${GRAFT_REPO_ROOT:-/root/repo}"
cd $R
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > gpurun_out/r3_gpu_tests.log 2>&1; echo rc=$?; tail -3 gpurun_out/r3_gpu_tests.log | cut -c1-220
