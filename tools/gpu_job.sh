#!/bin/bash
R="${GRAFT_REPO_ROOT:-/root/repo}"
cd $R
for cfg in "A=1" "MG_WGRAD=0" "MG_WGRAD=0 MG_BLUR_EPILOGUE=0" "MG_WGRAD=0 MG_MODCONV=0" "A=2"; do
  echo "== $cfg"
  env $cfg timeout -k 10 300 python -m pytest tests/test_train_step_gpu.py -q -m gpu -k "recorded_phases_equal" -s 2>&1 | grep -E "recorded-vs-eager|passed|failed" | cut -c1-200
done
