#!/usr/bin/env python3
"""Generates dlsa_amd/csrc/gram_mid_asm.inc: the per-wave tile plans of the mid-width Gram kernel (gram_mid.hip) --
NT = 8 .. 17 full 16-column tiles (+ G four-column tail groups), i.e. 125 <= p <= 284.

The NT (NT + 1) / 2 upper-triangle tiles are dealt to the 8 waves of a workgroup (two per SIMD, <= 20 tiles = 160 AGPRs
each): tile rows are taken in bands of 4, a band is walked column by column (a column of a band = up to 4 tiles sharing one
B fragment), and the walk is cut into 8 runs of equal length.  A wave's tiles then share few fragments (typically 4-5 A
+ 5-8 B per ~19 tiles).  Because every wave role has its own static tile list, everything is compile-time: the LDS read
offsets are immediates, the accumulators are named AGPRs, the epilogue knows where each tile goes.
The NT + 1 tail tiles (tile row t x the 4 G tail columns, v_mfma_f64_4x4x4_4b_f64, see gram_narrow.hip) go to waves that
already hold fragment t where possible.

usage: python3 tools/gen_gram_mid_asm.py > dlsa_amd/csrc/gram_mid_asm.inc"""
NT_MIN, NT_MAX, MAXG, NWAVES, BAND = 8, 17, 3, 8, 4


def plan(nt, g):
    """-> list of 8 roles: dict(tiles=[(ti, tj)], tails=[(t, gi)], rows=[A fragment tiles], cols=[B fragment tiles])"""
    walk = []
    for b0 in range(0, nt, BAND):
        rows = list(range(b0, min(b0 + BAND, nt)))
        for tj in range(b0, nt):
            col = [(ti, tj) for ti in rows if ti <= tj]
            walk.extend(col)
    n = len(walk)
    roles = []
    pos = 0
    for w in range(NWAVES):
        take = n // NWAVES + (1 if w < n % NWAVES else 0)
        tiles = walk[pos:pos + take]
        pos += take
        roles.append({"tiles": tiles, "tails": []})
    # tails: tile rows 0 .. nt (nt = the partial tile holding the tail columns themselves) x g groups
    if g:
        load = [len(r["tiles"]) * 4 for r in roles]            # in quarter-tiles
        cap = (nt + 1 + NWAVES - 1) // NWAVES                    # tail rows per wave
        nrows_t = [0] * NWAVES
        for t in range(nt + 1):
            holders = [w for w, r in enumerate(roles) if nrows_t[w] < cap and any(ti == t for ti, _ in r["tiles"])]
            cand = holders if holders else [w for w in range(NWAVES) if nrows_t[w] < cap]
            w = min(cand, key=lambda q: load[q])
            for gi in range(g):
                roles[w]["tails"].append((t, gi))
            load[w] += g
            nrows_t[w] += 1
    for r in roles:
        r["rows"] = sorted(set([ti for ti, _ in r["tiles"]] + [t for t, _ in r["tails"]]))
        r["cols"] = sorted(set(tj for _, tj in r["tiles"]))
    return roles


def emit(nt, g, out):
    roles = plan(nt, g)
    maxa = max(len(r["rows"]) for r in roles)
    maxb = max(max(len(r["cols"]) for r in roles), 1)
    maxt = max(len(r["tiles"]) for r in roles)
    maxs = max(len(r["tails"]) for r in roles)
    nreg = max(maxt * 8 + maxs * 2, 8)
    assert nreg <= 180, (nt, g, maxt, maxs)
    ga = max(g, 1)
    out.append("// ---- NT = %d, G = %d: tiles per wave %s, tails %s, A frags %s, B frags %s" % (
        nt, g, [len(r["tiles"]) for r in roles], [len(r["tails"]) for r in roles], [len(r["rows"]) for r in roles],
        [len(r["cols"]) for r in roles]))
    out.append("template <> struct MidPlan<%d, %d> {" % (nt, g))
    out.append("    static constexpr int MAXA = %d, MAXB = %d, NREG = %d;" % (maxa, maxb, nreg))
    # loads: fa[i] = fragment of tile rows[i], fb[j] = fragment of tile cols[j] (byte offset of tile t = 128 t)
    out.append("    template <int W, typename LD> static __device__ __forceinline__ void load(LD&& ld, double (&fa)[MAXA], double (&fb)[MAXB]) {")
    for w, r in enumerate(roles):
        stm = ["fa[%d] = ld(%d);" % (i, 128 * t) for i, t in enumerate(r["rows"])] + ["fb[%d] = ld(%d);" % (j, 128 * t) for j, t in enumerate(r["cols"])]
        out.append("        %sif constexpr (W == %d) { %s }" % ("" if w == 0 else "else ", w, " ".join(stm)))
    out.append("    }")
    out.append("    template <int W> static constexpr bool has_tails() { constexpr bool n[8] = {%s}; return n[W]; }" % ", ".join("true" if r["tails"] else "false" for r in roles))
    # weights: on the side with fewer fragments (tails ride on the A side: if B is scaled, bt is scaled too)
    out.append("    template <int W> static __device__ __forceinline__ void scale(double w, double (&fa)[MAXA], double (&fb)[MAXB], double (&bt)[%d]) {" % ga)
    for w, r in enumerate(roles):
        if len(r["rows"]) <= len(r["cols"]):
            stm = ["fa[%d] *= w;" % i for i in range(len(r["rows"]))]
        else:
            stm = ["fb[%d] *= w;" % j for j in range(len(r["cols"]))] + (["bt[%d] *= w;" % gi for gi in range(g)] if r["tails"] else [])
        out.append("        %sif constexpr (W == %d) { %s }" % ("" if w == 0 else "else ", w, " ".join(stm)))
    out.append("    }")
    out.append("    template <int W> static __device__ __forceinline__ void mfma(const double (&fa)[MAXA], const double (&fb)[MAXB], const double (&bt)[%d]) {" % ga)
    for w, r in enumerate(roles):
        lines = ["s_nop 1"]
        for k, (ti, tj) in enumerate(r["tiles"]):
            lines.append("v_mfma_f64_16x16x4_f64 a[%d:%d], %%%d, %%%d, a[%d:%d]" % (8 * k, 8 * k + 7, r["rows"].index(ti), maxa + r["cols"].index(tj), 8 * k, 8 * k + 7))
        base = 8 * len(r["tiles"])
        for k, (t, gi) in enumerate(r["tails"]):
            lines.append("v_mfma_f64_4x4x4_4b_f64 a[%d:%d], %%%d, %%%d, a[%d:%d]" % (base + 2 * k, base + 2 * k + 1, r["rows"].index(t), maxa + maxb + gi, base + 2 * k, base + 2 * k + 1))
        ops = ['"v"(fa[%d])' % i for i in range(maxa)] + ['"v"(fb[%d])' % j for j in range(maxb)] + ['"v"(bt[%d])' % gi for gi in range(ga)]
        clob = ", ".join('"a%d"' % q for q in range(base + 2 * len(r["tails"])))
        out.append('        %sif constexpr (W == %d) asm volatile("%s" :: %s : %s);' % ("" if w == 0 else "else ", w, "\\n\\t".join(lines), ", ".join(ops), clob))
    out.append("    }")
    out.append("    template <int W> static __device__ __forceinline__ void store(int lane, double* __restrict__ P, int PP) {")
    for w, r in enumerate(roles):
        stm = ["mid_store_tile<%d>(P, PP, lane, %d, %d);" % (k, ti, tj) for k, (ti, tj) in enumerate(r["tiles"])]
        base = 8 * len(r["tiles"])
        stm += ["mid_store_tail<%d>(P, PP, lane, %d, %d);" % (base + 2 * k, t, 16 * nt + 4 * gi) for k, (t, gi) in enumerate(r["tails"])]
        out.append("        %sif constexpr (W == %d) { %s }" % ("" if w == 0 else "else ", w, " ".join(stm)))
    out.append("    }")
    out.append("};")


def main():
    out = ["// GENERATED by tools/gen_gram_mid_asm.py -- do not edit.",
           "template <int NT, int G> struct MidPlan;",
           "template <int NREG> __device__ __forceinline__ void mid_acc_zero();"]
    sizes = set()
    for nt in range(NT_MIN, NT_MAX + 1):
        for g in range(MAXG + 1):
            roles = plan(nt, g)
            sizes.add(max(max(len(r["tiles"]) for r in roles) * 8 + max(len(r["tails"]) for r in roles) * 2, 8))
    for n in sorted(sizes):
        body = "\\n\\t".join("v_accvgpr_write_b32 a%d, 0" % r for r in range(n))
        out.append("template <> __device__ __forceinline__ void mid_acc_zero<%d>() {" % n)
        out.append('    asm volatile("%s" ::: %s);' % (body, ", ".join('"a%d"' % q for q in range(n))))
        out.append("}")
    # accumulator reads
    out.append("template <int K> __device__ __forceinline__ void mid_tile_read(double (&v)[4]) {")
    out.append("    int w0, w1, w2, w3, w4, w5, w6, w7;")
    for t in range(22):
        b = "\\n\\t".join("v_accvgpr_read_b32 %%%d, a%d" % (r, 8 * t + r) for r in range(8))
        out.append('    %sif constexpr (K == %d) asm volatile("%s" : "=v"(w0), "=v"(w1), "=v"(w2), "=v"(w3), "=v"(w4), "=v"(w5), "=v"(w6), "=v"(w7));'
                   % ("" if t == 0 else "else ", t, b))
    out.append("    v[0] = __hiloint2double(w1, w0); v[1] = __hiloint2double(w3, w2);")
    out.append("    v[2] = __hiloint2double(w5, w4); v[3] = __hiloint2double(w7, w6);")
    out.append("}")
    out.append("template <int R> __device__ __forceinline__ double mid_pair_read() {")
    out.append("    int lo, hi;")
    first = True
    for r in range(0, 180, 2):
        out.append('    %sif constexpr (R == %d) asm volatile("v_accvgpr_read_b32 %%0, a%d\\n\\tv_accvgpr_read_b32 %%1, a%d" : "=v"(lo), "=v"(hi));'
                   % ("" if first else "else ", r, r, r + 1))
        first = False
    out.append("    return __hiloint2double(hi, lo);")
    out.append("}")
    out.append("// tile K of the wave = H tile (ti, tj), ti <= tj: C/D register q of lane l = C[4q + (l >> 4)][l & 15]")
    out.append("template <int K> __device__ __forceinline__ void mid_store_tile(double* __restrict__ P, int PP, int lane, int ti, int tj) {")
    out.append("    double v[4];")
    out.append("    mid_tile_read<K>(v);")
    out.append("#pragma unroll")
    out.append("    for (int q = 0; q < 4; ++q) P[(int64_t)(16 * ti + 4 * q + (lane >> 4)) * PP + 16 * tj + (lane & 15)] = v[q];")
    out.append("}")
    out.append("// tail accumulator at AGPR R = tile row t x tail columns c0 .. c0 + 3: lane l holds H[16 t + 4 b + i][c0 + j], i = l >> 4, b = (l & 15) >> 2, j = l & 3")
    out.append("template <int R> __device__ __forceinline__ void mid_store_tail(double* __restrict__ P, int PP, int lane, int t, int c0) {")
    out.append("    P[(int64_t)(16 * t + 4 * ((lane & 15) >> 2) + (lane >> 4)) * PP + c0 + (lane & 3)] = mid_pair_read<R>();")
    out.append("}")
    for nt in range(NT_MIN, NT_MAX + 1):
        for g in range(MAXG + 1):
            emit(nt, g, out)
    print("\n".join(out))


if __name__ == "__main__":
    main()
