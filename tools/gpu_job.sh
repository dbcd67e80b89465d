#!/bin/bash
set -o pipefail
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out
timeout -k 10 500 python tools/debug_graph2.py > $O/r2_debug_graph2.log 2>&1
echo "debug rc=$?"; grep -v "^$" $O/r2_debug_graph2.log | tail -30
timeout -k 10 500 python tools/debug_graph2.py --no-share > $O/r2_debug_graph2b.log 2>&1
echo "debug noshare rc=$?"; grep -v "^$" $O/r2_debug_graph2b.log | tail -30
timeout -k 10 500 python tools/debug_graph2.py --ada > $O/r2_debug_graph2c.log 2>&1
echo "debug ada rc=$?"; grep -v "^$" $O/r2_debug_graph2c.log | tail -30
