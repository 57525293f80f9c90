#!/bin/bash
# A/B LARS builds on one box: bench/try_lars.sh <suffix> ...   (build/var/libdlsa_lars<suffix>.so)
cp dlsa_amd/libdlsa_hip.so /tmp/libdlsa_orig.so
echo "== in-tree"; python bench/lars_quick.py 2>&1 | grep -v amdgpu.ids
for v in "$@"; do cp build/var/libdlsa_lars$v.so dlsa_amd/libdlsa_hip.so; echo "== $v"; python bench/lars_quick.py 2>&1 | grep -v "amdgpu.ids\|p=50 \|p=50:"; done
cp /tmp/libdlsa_orig.so dlsa_amd/libdlsa_hip.so
