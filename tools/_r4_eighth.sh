mkdir -p gpurun_out/r4
for d in 0 256 512 1024 0 256 512 1024; do SR_GEMM_DEBUG=$d timeout -k 10 120 python tools/conv_time.py 6144 2>&1 | tail -1 | cut -c1-120; done | tee gpurun_out/r4/prio_ab.txt
for d in 4 260 516 1028; do echo "DEBUG=$d"; SR_LIB_PATH=$PWD/situation_recognition_amd/libsrhip_stamps.so SR_GEMM_DEBUG=$d timeout -k 10 200 python3 tools/stamp_layer3.py 2>&1 | grep -v Warn | tail -3; done | tee gpurun_out/r4/stamps_ab.txt
