#!/bin/bash
# FETCH_SIZE and WRITE_SIZE in SEPARATE rocprofv3 passes (kernel-trace only), as MI355X_MICROARCH.md prescribes
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03_pmc; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for cfg in "upfirdn2d bfloat16 nhwc" "upfirdn2d128 bfloat16 nhwc" "bwd_fused bfloat16 nhwc" "epilogue bfloat16 nhwc" "wgrad bfloat16 nhwc" "torgb bfloat16 nhwc" "modconv bfloat16 nhwc" "bias_act bfloat16 nhwc"; do
  set -- $cfg; k=$1; dt=$2; lay=${3:-nchw}
  for ctr in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/pmc_run
    timeout -k 10 200 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d /tmp/pmc_run -o run -- python3 $R/tools/kernel_loop.py $k $dt 3 $lay > /tmp/pmc_run.log 2>&1
    f=$(find /tmp/pmc_run -name "*counter_collection.csv" | head -1)
    [ -n "$f" ] && cp "$f" $O/${k}_${dt}_${lay}_${ctr}_counter_collection.csv
    tail -1 /tmp/pmc_run.log
  done
done
ls -la $O | head -30
