#!/bin/bash
R="${GRAFT_REPO_ROOT:-/root/repo}"
cd $R
timeout -k 10 600 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --conv-table gpurun_out/r03_conv_table_b32.csv > gpurun_out/r3_bench_ct.json 2> gpurun_out/r3_bench_ct.err; echo rc=$?
head -45 gpurun_out/r03_conv_table_b32.csv | cut -c1-200
