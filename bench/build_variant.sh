#!/bin/bash
# builds build/var/libdlsa_<name>.so: the product objects with ONE source recompiled under extra flags
#   bench/build_variant.sh <name> <source under dlsa_amd/csrc> [flags...]      e.g.  bench/build_variant.sh wgs2 gram_narrow.hip -DDLSA_NARROW_WGS2_MAXREG=184
name=$1; src=$2; shift 2
mkdir -p build/var
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Iinclude -Wno-unused-function "$@" -x hip -c dlsa_amd/csrc/$src -o build/var/${src}_$name.o || exit 1
hipcc --offload-arch=gfx950 -shared -fPIC $(ls build/*.o | grep -v "build/$src.o") build/var/${src}_$name.o -ldl -o build/var/libdlsa_$name.so
