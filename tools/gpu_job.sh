#!/bin/bash
R="${GRAFT_REPO_ROOT:-/root/repo}"
cd $R
O=$R/gpurun_out
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/r2_smoke.log 2>&1
echo "smoke rc=$?"; tail -1 $O/r2_smoke.log
timeout -k 10 400 python bench.py --workload single --no-cpu-baseline > $O/r02_bench_single_b32.json 2> $O/r02_bench_single_b32.err
echo "single rc=$?"; python -c "
import json,sys
d=json.loads(open('$O/r02_bench_single_b32.json').read().strip().splitlines()[-1]); print(d['metric'], d['ms_per_step'], d['value'], d['roofline']['frac'] if d.get('roofline') else None, d['roofline_mfma']['frac'] if d.get('roofline_mfma') else None, d['config'].get('phase_graphs'))"
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > $O/r2_gpu_tests_final.log 2>&1
rc=$?; echo "gpu tests rc=$rc"; tail -3 $O/r2_gpu_tests_final.log
