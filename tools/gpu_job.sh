#!/bin/bash
R="${GRAFT_REPO_ROOT:-/root/repo}"
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof && mkdir -p /tmp/prof
cd $R
timeout -k 10 900 rocprofv3 --kernel-trace --stats -d /tmp/prof -o r3 --output-format csv -- python3 bench.py --steps 16 --warmup 4 --no-cpu-baseline --prof-steps 0 --marker-file /tmp/prof/marker.json > gpurun_out/r3_trace_b32.json 2> gpurun_out/r3_trace_b32.err; echo rc=$?
T=$(find /tmp/prof -name "*kernel_trace.csv" | head -1); S=$(find /tmp/prof -name "*kernel_stats.csv" | head -1)
python tools/summarize_trace.py $T /tmp/prof/marker.json gpurun_out/r03_aio_b32_timed_region_kernels.csv | tail -2
cp $S gpurun_out/r03_aio_b32_rocprofv3_kernel_stats.csv
