#!/bin/bash
# same-box A/B of an SR_GEMM_DEBUG switch: alternating processes
R=$GRAFT_REPO_ROOT; [ -n "$R" ] || R=$PWD
O=$R/gpurun_out/ab; mkdir -p $O; cd $R
for r in 1 2 3; do for d in ${AB:-0 64}; do SR_GEMM_DEBUG=$d timeout -k 10 120 python3 tools/conv_time.py 2>/dev/null | tee -a $O/ab.txt; done; done
