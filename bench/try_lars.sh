#!/bin/bash
cp dlsa_amd/libdlsa_hip.so /tmp/libdlsa_orig.so
echo "== 1024"; python bench/lars_quick.py
for v in "$@"; do cp build/var/libdlsa_lars$v.so dlsa_amd/libdlsa_hip.so; echo "== $v"; python bench/lars_quick.py; done
cp /tmp/libdlsa_orig.so dlsa_amd/libdlsa_hip.so
