#!/bin/bash
R="${GRAFT_REPO_ROOT:-/root/repo}"
O=$R/gpurun_out
cd $R
timeout -k 10 1000 python3 -m pytest tests -m gpu -x -q > $O/pytest_gpu_r3b.log 2>&1; echo "pytest rc=$?"
tail -6 $O/pytest_gpu_r3b.log | cut -c1-300
timeout -k 10 300 python3 bench.py --steps 32 --warmup 8 > $O/r3_bench_b32_mid.json 2> $O/r3_bench_b32_mid.err; echo "bench rc=$?"
python3 -c "
import json;d=json.loads(open('$O/r3_bench_b32_mid.json').read().strip().splitlines()[-1]);print(d['ms_per_step'],d['value'],d['roofline']['kernel'],d['roofline']['frac'],d['roofline_mfma']['frac'])"
