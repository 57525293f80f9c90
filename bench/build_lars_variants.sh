#!/bin/bash
# builds build/var/libdlsa_lars<T>_<TRIP>.so for "threads:trip" pairs, e.g. bench/build_lars_variants.sh 1024:8 512:32
mkdir -p build/var
for v in "$@"; do
  T=${v%%:*}; R=${v##*:}
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Iinclude -DDLSA_LARS_THREADS=$T -DDLSA_LARS_TRIP=$R -DDLSA_LARS_PROF -x hip -c dlsa_amd/csrc/lars.hip -o build/var/lars_${T}_${R}.o -Rpass-analysis=kernel-resource-usage 2>&1 | grep -A12 "lars_kernel" | grep "VGPRs:\|Scratch" | tr '\n' ' '; echo " <- $v"
  hipcc --offload-arch=gfx950 -shared -fPIC $(ls build/*.o | grep -v lars.hip.o) build/var/lars_${T}_${R}.o -o build/var/libdlsa_lars${T}_${R}.so
done
