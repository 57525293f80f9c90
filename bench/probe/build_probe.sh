#!/bin/bash
# bench/probe/build_probe.sh <tag> <NT> [-DPLAN_PROBE_... flags]  ->  build/probe/probe_<tag>   (one width, HASW, every G)
tag=$1; nt=$2; shift 2
mkdir -p build/probe build/gen
C=$(python3 -c "import sys; sys.path.insert(0,'tools'); import gen_gram_plan_asm as g; print(g.groups_for($nt))")
[ -f build/gen/gram_plan_common.inc ] || python3 tools/gen_gram_plan_asm.py common > build/gen/gram_plan_common.inc
python3 tools/gen_gram_plan_asm.py plans $C $nt $nt > build/probe/gram_plan_${C}_${nt}_${nt}.inc
F="--offload-arch=gfx950 -O3 -std=c++17 -Iinclude -Idlsa_amd/csrc -Ibuild/gen -Ibuild/probe -Wno-inline-asm -Wno-unused-function -Wno-unused-result"
hipcc $F -DPLAN_PROBE -DPLAN_LO=$nt -DPLAN_HI=$nt -DPLAN_INC="\"gram_plan_${C}_${nt}_${nt}.inc\"" "$@" -x hip -c dlsa_amd/csrc/gram_plan_unit.hip -o build/probe/unit_$tag.o || exit 1
hipcc $F -DPROBE_NT=$nt -DPROBE_TAG="\"$tag\"" -x hip -c bench/probe/plan_probe_main.hip -o build/probe/main_$tag.o || exit 1
[ -f build/probe/error.o ] || hipcc $F -x hip -c dlsa_amd/csrc/error.cpp -o build/probe/error.o
hipcc --offload-arch=gfx950 build/probe/unit_$tag.o build/probe/main_$tag.o build/probe/error.o -o build/probe/probe_$tag
