#!/bin/bash
R="${GRAFT_REPO_ROOT:-/root/repo}"
cd $R
timeout -k 10 200 python tools/debug_epilogue.py > gpurun_out/debug_epilogue.log 2>&1; echo rc=$?; grep -c "mismatches 0" gpurun_out/debug_epilogue.log; grep -v "mismatches 0" gpurun_out/debug_epilogue.log | tail -5
timeout -k 10 600 python -m pytest tests/test_hip_parity.py -x -q -m gpu > gpurun_out/wg_tests.log 2>&1; echo rc=$?; tail -5 gpurun_out/wg_tests.log | cut -c1-220
