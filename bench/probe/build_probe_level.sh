#!/bin/bash
# like build_probe.sh, with GP_LEVEL=1 plans (every role padded to the heaviest role's LDS reads and multiplications)
tag=$1; nt=$2; shift 2
C=$(python3 -c "import sys; sys.path.insert(0,'tools'); import gen_gram_plan_asm as g; print(g.groups_for($nt))")
mkdir -p build/probe/level
GP_LEVEL=1 python3 tools/gen_gram_plan_asm.py plans $C $nt $nt > build/probe/level/gram_plan_${C}_${nt}_${nt}.inc
F="--offload-arch=gfx950 -O3 -std=c++17 -Iinclude -Idlsa_amd/csrc -Ibuild/gen -Ibuild/probe/level -Wno-inline-asm -Wno-unused-function -Wno-unused-result"
hipcc $F -DPLAN_PROBE -DPLAN_LO=$nt -DPLAN_HI=$nt -DPLAN_INC="\"gram_plan_${C}_${nt}_${nt}.inc\"" "$@" -x hip -c dlsa_amd/csrc/gram_plan_unit.hip -o build/probe/unit_$tag.o || exit 1
hipcc $F -DPROBE_NT=$nt -DPROBE_TAG="\"$tag\"" -x hip -c bench/probe/plan_probe_main.hip -o build/probe/main_$tag.o || exit 1
[ -f build/probe/error.o ] || hipcc $F -x hip -c dlsa_amd/csrc/error.cpp -o build/probe/error.o
hipcc --offload-arch=gfx950 build/probe/unit_$tag.o build/probe/main_$tag.o build/probe/error.o -o build/probe/probe_$tag
