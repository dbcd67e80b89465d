#!/bin/bash
# rocprofv3 kernel trace of `bench.py --batch-gpu 4` (phases replayed from HIP graphs), cut to the timed region
R="${GRAFT_REPO_ROOT:-/root/repo}"
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/trace_run
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/trace_run -o run -- python3 $R/bench.py --batch-gpu 4 --steps 4 --warmup 20 --no-cpu-baseline --prof-steps 0 --marker-file /tmp/marker.json > $O/r2_trace_b4.json 2> $O/r2_trace_b4.err
echo "trace rc=$?"
T=$(ls /tmp/trace_run/*kernel_trace.csv /tmp/trace_run/*/*kernel_trace.csv 2>/dev/null | head -1)
S=$(ls /tmp/trace_run/*kernel_stats.csv /tmp/trace_run/*/*kernel_stats.csv 2>/dev/null | head -1)
python3 $R/tools/summarize_trace.py $T /tmp/marker.json $O/r02_aio_b4_timed_region_kernels.csv > /dev/null && head -c 60000 $S > $O/r02_aio_b4_rocprofv3_kernel_stats.csv
echo "summary rc=$?"; head -1 $O/r02_aio_b4_timed_region_kernels.csv
python3 -c "
import json
d=json.loads(open('$O/r2_trace_b4.json').read().strip().splitlines()[-1]); print(d['ms_per_step'], d['config'].get('phase_graphs',{}).get('recorded_and_verified'))"
