"""CPU: post-build checks of the shipped library's code objects (no GPU needed).

The AGPR-resident Gram kernels (gram_narrow / gram_plan / gram_cyclic) keep loop-carried accumulators in named AGPRs that
their inline-asm blocks only list as clobbers; tools/check_agpr_kernels.py verifies on the built libdlsa_hip.so that the
compiler left those registers alone: no scratch, the planned AGPR counts, VGPR + AGPR within two waves per SIMD, and no
v_accvgpr moves inside any loop of the kernels' control-flow graphs."""
import os
import shutil
import sys

import pytest

from conftest import ROOT

sys.path.insert(0, os.path.join(ROOT, "tools"))


def test_agpr_resident_kernels_are_untouched_by_the_compiler():
    import check_agpr_kernels as chk
    lib = os.path.join(ROOT, "dlsa_amd", "libdlsa_hip.so")
    if not os.path.exists(lib):
        pytest.fail("libdlsa_hip.so has not been built (run `make`)")
    if not (os.path.exists(chk.tool("llvm-objdump")) or shutil.which("llvm-objdump")):
        pytest.skip("llvm-objdump not available")
    n, problems = chk.check(lib)
    # 2 weights x (narrow: NT 3..6 x G 0..3 + NT 7) + cyclic 2 x 4 + the plan kernels (weighted only)
    assert n >= 140, "only %d AGPR-resident kernels found in the library" % n
    assert not problems, "\n".join("%s: %s" % (k, "; ".join(e)) for k, e in problems)


def test_loop_detection_flags_a_move_inside_a_loop():
    """The checker's CFG / strongly-connected-component pass on a hand-made instruction list."""
    import check_agpr_kernels as chk
    ins = [(0, "s_mov_b32", None), (4, "v_accvgpr_write_b32", None),            # prologue
           (8, "v_mfma_f64_16x16x4_f64", None), (16, "v_accvgpr_read_b32", None), (20, "s_cbranch_scc1", 8),   # loop 8..20
           (24, "v_accvgpr_read_b32", None), (28, "s_endpgm", None)]            # epilogue
    inloop, nblocks = chk._cyclic_blocks(ins)
    assert nblocks == 1 and inloop == {8, 16, 20}
    # a forward jump over a block and a backward JUMP that is not a cycle (layout artefact) are not loops
    ins2 = [(0, "s_branch", 12), (4, "v_accvgpr_read_b32", None), (8, "s_endpgm", None), (12, "s_nop", None), (16, "s_branch", 4)]
    inloop2, nblocks2 = chk._cyclic_blocks(ins2)
    assert nblocks2 == 0 and not inloop2
