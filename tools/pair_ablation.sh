#!/bin/bash
# Diagnostic builds of csrc/pair.hip with parts compiled out (PAIR_ABL bits: 1 MFMAs, 2 HBM traffic, 4 weight LDS-DMA, 8 Z epilogue
# arithmetic, 16 statistics reduction), each timed at batch 6144 on the same box.  usage: tools/pair_ablation.sh "0 1 2 4 8 16 ..." [batch]
set -e
cd "$(dirname "$0")/.."
C=situation_recognition_amd/csrc
FL="--offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wno-unused-value -mllvm -pragma-unroll-threshold=400000"
B=${2:-6144}
for a in $1; do
  /opt/rocm/bin/hipcc $FL -DPAIR_ABL=$a -c $C/pair.hip -o $C/_obj/pair_abl$a.o
  objs=$(ls $C/_obj/*.o | grep -v "_stamps\|pair_abl\|/pair.o\|amdgcn")
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o situation_recognition_amd/libsrhip_abl$a.so $objs $C/_obj/pair_abl$a.o -ldl
  echo "== PAIR_ABL=$a"
  SR_LIB_PATH=$PWD/situation_recognition_amd/libsrhip_abl$a.so python tools/pair_time.py $B time | tail -2
done
