#!/bin/bash
# builds build/var/libdlsa_larsq_prof.so: the library with lars_q.hip's phase timer (-DDLSA_LARS_PROF); run bench/lars_prof.py on it
mkdir -p build/var
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Iinclude -DDLSA_LARS_PROF $* -x hip -c dlsa_amd/csrc/lars_q.hip -o build/var/lars_q_prof.o || exit 1
hipcc --offload-arch=gfx950 -shared -fPIC $(ls build/*.o | grep -v lars_q.hip.o) build/var/lars_q_prof.o -ldl -o build/var/libdlsa_larsq_prof.so
