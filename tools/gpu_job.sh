#!/bin/bash
R="${GRAFT_REPO_ROOT:-/root/repo}"
cd $R
O=$R/gpurun_out
timeout -k 10 600 python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "full_size or sep12 or reflect or warp" > $O/r2_parity.log 2>&1
rc=$?; echo "parity rc=$rc"; tail -12 $O/r2_parity.log
