#!/bin/bash
# An A/B build of libavrf.so that differs from the shipped one in the compile flags of a few units:
#   bash tools/build_variant.sh NAME "-DFLAG ..." [unit ...]      ->  build/variants/libavrf_NAME.so
# unit: msm (default), capi, ring, ... or vrf_batch:K / vrf_single:K for the per-suite units (K = suite id).  The other objects are
# taken from build/obj (run the shipped build first).  Use with AVRF_LIB_PATH / tools/ab.sh lib=PATH.
set -e
cd "$(dirname "$0")/../ark_vrf_amd/csrc"
NAME=$1; FLAGS=$2; shift 2; UNITS=${*:-msm}
mkdir -p ../../build/variants ../../build/obj_$NAME
CXX="/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Xarch_host -march=x86-64-v3 -mllvm -enable-ipra=0 -Wno-unused-function -Wno-unused-result -Wno-pass-failed"
SKIP=""
for u in $UNITS; do
  base=${u%%:*}; k=${u#*:}
  if [ "$base" != "$u" ]; then obj=${base}_s$k.o; $CXX $FLAGS -DAVRF_TU_SUITE=$k -c -o ../../build/obj_$NAME/$obj $base.hip
  else obj=$base.o; $CXX $FLAGS -c -o ../../build/obj_$NAME/$obj $base.hip; fi
  SKIP="$SKIP /$obj"
done
OBJS=""
for o in ../../build/obj/*.o; do keep=1; for sk in $SKIP; do case $o in *$sk) keep=0;; esac; done; [ $keep = 1 ] && OBJS="$OBJS $o"; done
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o ../../build/variants/libavrf_$NAME.so ../../build/obj_$NAME/*.o $OBJS -lpthread
echo built build/variants/libavrf_$NAME.so
