#!/bin/bash
# A/B of two source variants of the pair kernel on one box: tools/pair_ab.sh "v1 v2" [batch]  (csrc/pair_<name>.hip)
set -e
cd "$(dirname "$0")/.."
C=situation_recognition_amd/csrc
FL="--offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wno-unused-value -mllvm -pragma-unroll-threshold=400000"
B=${2:-6144}
for a in $1; do
  /opt/rocm/bin/hipcc $FL ${PAIR_EXTRA} -c $C/pair_$a.hip -o $C/_obj/pair_ab_$a.o
  objs=$(ls $C/_obj/*.o | grep -v "_stamps\|pair_abl\|pair_st\|pair_ab_\|/pair.o\|amdgcn")
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o situation_recognition_amd/libsrhip_ab_$a.so $objs $C/_obj/pair_ab_$a.o -ldl
done
for rep in 1 2; do for a in $1; do
  echo "== variant $a"
  SR_LIB_PATH=$PWD/situation_recognition_amd/libsrhip_ab_$a.so python tools/pair_time.py $B time | tail -1
done; done
