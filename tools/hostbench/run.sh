#!/bin/bash
# host-only: the multi-buffer SHA-512 variants of csrc/host_sha512_mb.h on this box's CPU (no GPU work)
cd "$(dirname "$0")"
grep -m1 "model name" /proc/cpuinfo
for cc in "g++ -O3 -march=x86-64-v3" "g++ -O3 -march=native" "/opt/rocm/lib/llvm/bin/clang++ -O3 -march=x86-64-v3" "/opt/rocm/lib/llvm/bin/clang++ -O3 -march=native"; do
  echo "== $cc"; $cc -I../../ark_vrf_amd/csrc -o /tmp/hb_bin hb.cpp 2>&1 | head -3 && /tmp/hb_bin
done
