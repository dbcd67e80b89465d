#!/bin/bash
R="${GRAFT_REPO_ROOT:-/root/repo}"
cd $R
timeout -k 10 500 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3 | cut -c1-300
timeout -k 10 300 python bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | python -c "import sys, json; d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('driver-style', d['ms_per_step'], d['value'], d['n_gpus'], d['steps'], d['warmup'])"
