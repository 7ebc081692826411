#!/bin/bash
R=$GRAFT_REPO_ROOT; [ -n "$R" ] || R=$PWD
O=$R/gpurun_out/stamps; mkdir -p $O; cd $R
SR_LIB_PATH=$R/situation_recognition_amd/libsrhip_stamps.so SR_GEMM_DEBUG=4 timeout -k 10 200 python3 tools/stamp_layer3.py > $O/layer3.txt 2>&1; cat $O/layer3.txt
