#!/bin/bash
R="${GRAFT_REPO_ROOT:-/root/repo}"
cd $R
timeout -k 10 300 python tools/microbench.py --layout nhwc --batch 32 --dtype bfloat16 2>&1 | grep -E "upfirdn2d" | cut -c1-100
timeout -k 10 400 python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "upfirdn or blur or nhwc" 2>&1 | tail -3 | cut -c1-200
