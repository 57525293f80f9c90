#!/bin/bash
# builds build/var/libdlsa_narrow<KC>.so for chunk sizes, e.g. bench/build_narrow_variants.sh 32 48
mkdir -p build/var
for kc in "$@"; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Iinclude -DDLSA_NARROW_KC=$kc -x hip -c dlsa_amd/csrc/gram_narrow.hip -o build/var/gram_narrow_$kc.o || exit 1
  hipcc --offload-arch=gfx950 -shared -fPIC $(ls build/*.o | grep -v gram_narrow.hip.o) build/var/gram_narrow_$kc.o -o build/var/libdlsa_narrow$kc.so
done
