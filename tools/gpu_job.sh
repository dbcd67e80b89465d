#!/bin/bash
R="${GRAFT_REPO_ROOT:-/root/repo}"
cd $R
timeout -k 10 400 python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "blur or epilogue or upfirdn" 2>&1 | tail -4 | cut -c1-250
timeout -k 10 200 python tools/bench_blur_epilogue.py 2>&1 | grep -E "fused=True|blur alone" | cut -c1-120
