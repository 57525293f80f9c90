#!/bin/bash
# build/var/libdlsa_f16.so: the image of sqrt(w) [X | 1] in fp16 instead of bf16 (logit.hip + irls_wide.hip under -DDLSA_IMG_F16=1)
mkdir -p build/var
for src in logit.hip irls_wide.hip; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Iinclude -Wno-unused-function -DDLSA_IMG_F16=1 -x hip -c dlsa_amd/csrc/$src -o build/var/${src}_f16.o || exit 1
done
hipcc --offload-arch=gfx950 -shared -fPIC $(ls build/*.o | grep -v "build/logit.hip.o" | grep -v "build/irls_wide.hip.o") build/var/logit.hip_f16.o build/var/irls_wide.hip_f16.o -ldl -o build/var/libdlsa_f16.so
