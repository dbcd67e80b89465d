#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out
mkdir -p $O/miopen_b816 && cp docker-montage-gan_amd/tuning/miopen/*.txt $O/miopen_b816/
( while true; do sleep 60; echo "[find] $(date +%T) $(wc -l $O/miopen_b816/*.ufdb.txt | tail -1)"; done ) &
HB=$!
timeout -k 10 560 python bench.py --batch-gpu 16 --graphs off --miopen find --miopen-db-out $O/miopen_b816 --steps 2 --warmup 1 --no-cpu-baseline --prof-steps 0 > $O/r2_find_b16.json 2> $O/r2_find_b16.err
echo "find b16 rc=$?"
timeout -k 10 560 python bench.py --batch-gpu 8 --graphs off --miopen find --miopen-db-out $O/miopen_b816 --steps 2 --warmup 1 --no-cpu-baseline --prof-steps 0 > $O/r2_find_b8.json 2> $O/r2_find_b8.err
echo "find b8 rc=$?"; kill $HB
wc -l $O/miopen_b816/*.txt
