#!/bin/bash
R="${GRAFT_REPO_ROOT:-/root/repo}"
cd $R
O=$R/gpurun_out
timeout -k 10 300 python tools/plugin_call_cost.py > $O/r02_plugin_call_cost.log 2>&1
echo "rc=$?"; cat $O/r02_plugin_call_cost.log | tail -5
