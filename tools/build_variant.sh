#!/bin/bash
# An A/B build of libavrf.so that differs from the shipped one in msm.hip's compile flags only (the bucket-accumulation
# policies are all in that unit):   bash tools/build_variant.sh NAME "-DFLAG ..."   ->  build/variants/libavrf_NAME.so
# (run the shipped build first: the other objects are taken from build/obj).  Use with AVRF_LIB_PATH / tools/ab.sh lib=PATH.
set -e
cd "$(dirname "$0")/../ark_vrf_amd/csrc"
NAME=$1; FLAGS=$2
mkdir -p ../../build/variants ../../build/obj_$NAME
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Xarch_host -march=x86-64-v3 -mllvm -enable-ipra=0 -Wno-unused-function -Wno-unused-result -Wno-pass-failed $FLAGS -c -o ../../build/obj_$NAME/msm.o msm.hip
OBJS=$(ls ../../build/obj/*.o | grep -v '/msm.o$')
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o ../../build/variants/libavrf_$NAME.so ../../build/obj_$NAME/msm.o $OBJS -lpthread
echo built build/variants/libavrf_$NAME.so
