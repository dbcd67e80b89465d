#!/bin/bash
R="${GRAFT_REPO_ROOT:-/root/repo}"
cd $R
timeout -k 10 300 python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "residual_tail" 2>&1 | grep -E "^E |Error|assert" | head -12 | cut -c1-300
