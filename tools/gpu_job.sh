#!/bin/bash
R="${GRAFT_REPO_ROOT:-/root/repo}"
cd $R
O=$R/gpurun_out
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > $O/r2_gpu_tests_final.log 2>&1
rc=$?; echo "gpu tests rc=$rc"; tail -3 $O/r2_gpu_tests_final.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 600 python bench.py > $O/r02_bench_b32_final.json 2> $O/r02_bench_b32_final.err
echo "b32 rc=$?"; python -c "
import json,sys
d=json.loads(open('$O/r02_bench_b32_final.json').read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'], d['cpu_baseline']['value'], d['roofline']['frac'], d['roofline_mfma']['frac'])"
for b in 16 8 4; do
timeout -k 10 300 python bench.py --batch-gpu $b --no-cpu-baseline --prof-steps 0 > $O/r02_bench_b${b}_final.json 2> $O/r02_bench_b${b}_final.err
echo "b$b rc=$?"; python -c "
import json,sys
d=json.loads(open('$O/r02_bench_b${b}_final.json').read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'], d['config'].get('phase_graphs',{}).get('recorded_and_verified'))"
done
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/trace_run
timeout -k 10 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/trace_run -o run -- python3 $R/bench.py --steps 4 --warmup 4 --no-cpu-baseline --prof-steps 0 --marker-file /tmp/marker.json > $O/r2_trace_bench.json 2> $O/r2_trace_bench.err
echo "trace rc=$?"
T=$(ls /tmp/trace_run/*kernel_trace.csv /tmp/trace_run/*/*kernel_trace.csv 2>/dev/null | head -1)
S=$(ls /tmp/trace_run/*kernel_stats.csv /tmp/trace_run/*/*kernel_stats.csv 2>/dev/null | head -1)
python3 $R/tools/summarize_trace.py $T /tmp/marker.json $O/r02_aio_b32_timed_region_kernels.csv > /dev/null && head -c 60000 $S > $O/r02_aio_b32_rocprofv3_kernel_stats.csv
echo "summary rc=$?"; head -1 $O/r02_aio_b32_timed_region_kernels.csv
