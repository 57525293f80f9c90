"""CPU: the generated tile plans of the plan-driven fp64 Gram kernel (tools/gen_gram_plan_asm.py) -- every upper-triangle
tile and every tail tile belongs to exactly one wave role, the accumulators fit, the SIMD loads of a workgroup group are
level, and the interleaved column layout maps every fragment position to a distinct column of H."""
import importlib.util
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("gen_gram_plan_asm", os.path.join(ROOT, "tools", "gen_gram_plan_asm.py"))
gen = importlib.util.module_from_spec(spec)
spec.loader.exec_module(gen)

SHAPES = [(nt, g) for nt in range(8, 36) for g in range(4)]


@pytest.mark.parametrize("nt,g", SHAPES)
def test_plan_covers_triangle_once(nt, g):
    C = gen.groups_for(nt)
    roles = gen.plan(nt, g, C)
    assert len(roles) == 8 * C
    tiles = [t for r in roles for t in r["tiles"]]
    assert sorted(tiles) == sorted((i, j) for i in range(nt) for j in range(i, nt))
    tails = [t for r in roles for t in r["tails"]]
    assert sorted(tails) == sorted((t, gi) for t in range(nt + 1) for gi in range(g))
    for r in roles:
        assert gen.role_regs(r) <= gen.MAX_AGPR
        for ti, tj in r["tiles"]:
            assert ti in r["a_set"] and tj in r["b_set"]
        # every fragment the role needs is fetched by exactly one LDS read
        got = sorted(i for _, _, idx in gen.loads_of(nt, r) for i in idx)
        assert got == list(range(len(r["frags"])))


@pytest.mark.parametrize("nt,g", SHAPES)
def test_plan_simd_loads_level(nt, g):
    C = gen.groups_for(nt)
    roles = gen.plan(nt, g, C)
    load = lambda m, s: sum(roles[m * 8 + s + 4 * h]["load"] + roles[m * 8 + s + 4 * h]["pad"] for h in range(2))
    tops = [max(load(m, s) for s in range(4)) for m in range(C)]
    assert len(set(tops)) == 1                                  # the workgroups of a group finish a k-step together
    total = sum(r["load"] for r in roles)
    assert total / (4.0 * C) / tops[0] > 0.93                   # SIMD balance


@pytest.mark.parametrize("nt", range(8, 36))
def test_interleaved_layout_is_a_permutation(nt):
    cols = []
    for t in range(nt):
        base, stride = gen.colmap(nt, t)
        cols += [base + stride * r for r in range(16)]
    assert sorted(cols) == list(range(16 * nt))
    assert gen.colmap(nt, nt) == (16 * nt, 1)                   # the tail tile is a plain block behind the full tiles


def test_group_sizes_match_the_kernel_header():
    src = open(os.path.join(ROOT, "dlsa_amd", "csrc", "gram_plan.h")).read()
    assert "nt <= 17 ? 1 : nt <= 24 ? 2 : 4" in src
    for nt in range(8, 36):
        assert gen.groups_for(nt) == (1 if nt <= 17 else 2 if nt <= 24 else 4)
