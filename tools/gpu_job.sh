#!/bin/bash
R="${GRAFT_REPO_ROOT:-/root/repo}"
cd $R
for cfg in "MG_FIR_CHUNK_BIG=32 MG_FIR_CHUNK_MID=16" "MG_FIR_CHUNK_BIG=64 MG_FIR_CHUNK_MID=32" "MG_FIR_CHUNK_BIG=128 MG_FIR_CHUNK_MID=64" "MG_FIR_CHUNK_BIG=16 MG_FIR_CHUNK_MID=8"; do
echo "== $cfg"
env $cfg timeout -k 10 300 python tools/microbench.py --layout nhwc --batch 32 --dtype bfloat16 2>&1 | grep -E "upfirdn2d 4x4 1:1 \[32,(64|128)" | cut -c1-100
done
