#!/bin/bash
R="${GRAFT_REPO_ROOT:-/root/repo}"
O=$R/gpurun_out
cd $R
timeout -k 10 400 python3 tools/graph_fault_probe.py --phase global_Gmain --replays 4 --top 0 --eager-repeats 4 --wrong-rng 1 > $O/probe_gmain_quant.log 2>&1; echo "gmain rc=$?"
timeout -k 10 400 python3 tools/graph_fault_probe.py --phase local_Gmain_layer1 --replays 3 --top 0 --eager-repeats 3 --wrong-rng 1 > $O/probe_local_quant.log 2>&1; echo "local rc=$?"
timeout -k 10 400 python3 tools/eager_spread_toy.py 6 > $O/eager_spread_toy.log 2>&1; echo "toy rc=$?"
