#!/bin/bash
# build/var/libdlsa_lstl.so: the product objects with irls_pass.hip and irls_batch.hip rebuilt under -DFP_TIMELINE=1 (bench/lockstep_timeline.py)
mkdir -p build/var
for src in irls_pass.hip irls_batch.hip; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Iinclude -Wno-unused-function -DFP_TIMELINE=1 -x hip -c dlsa_amd/csrc/$src -o build/var/${src}_lstl.o || exit 1
done
hipcc --offload-arch=gfx950 -shared -fPIC $(ls build/*.o | grep -v "build/irls_pass.hip.o" | grep -v "build/irls_batch.hip.o") build/var/irls_pass.hip_lstl.o build/var/irls_batch.hip_lstl.o -ldl -o build/var/libdlsa_lstl.so
