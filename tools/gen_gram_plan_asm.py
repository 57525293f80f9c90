#!/usr/bin/env python3
"""Generates the per-wave tile plans of the plan-driven fp64 Gram kernel (dlsa_amd/csrc/gram_plan_kernel.inc).

A design of NT full 16-column tiles (+ G four-column tail groups) has NT (NT + 1) / 2 upper-triangle tiles.  They are dealt
to the 8 C waves of a group of C workgroups (C = 1, 2, 4 CUs of one XCD; two waves per SIMD, <= 20 tiles = 160 AGPRs per
wave): tile rows are taken in bands of 4, a band is walked column by column, the walk is cut into 8 C runs of equal length.
The NT + 1 tail tiles (tile row t x the 4 G tail columns, v_mfma_f64_4x4x4_4b_f64) go to waves that already hold
fragment t where possible.  The runs are then paired onto SIMDs (heaviest with lightest) and the pairs dealt round-robin to
the workgroups of the group, so that every SIMD -- and with it every workgroup that shares its rows through L2 with the
others -- carries the same MFMA time to within one tile.

Everything about a wave role is static, so it is emitted as code: LDS read offsets are immediates, accumulators are named
AGPRs, the epilogue knows where each tile goes.  Column layout in LDS: tiles 2g and 2g + 1 are the EVEN and ODD columns
of the 32-column group g, so that ONE ds_read_b128 fetches the fragments of both (a fragment feeds the A side, the B side
or both); with NT odd the last full tile, and always the tail tile, are plain 16-column blocks.

usage: gen_gram_plan_asm.py common            > gram_plan_common.inc
       gen_gram_plan_asm.py plans C LO HI     > plans for NT = LO .. HI on C-workgroup groups
       gen_gram_plan_asm.py stats             (table of the plans)"""
import os
import sys

MAXG, BAND, MAX_AGPR, TAIL_REG_CAP = 3, 4, 184, 168


def groups_for(nt):
    """CUs per slab group for a width of nt full tiles (None: not served)."""
    if 8 <= nt <= 17:
        return 1
    if 18 <= nt <= 24:
        return 2
    if 25 <= nt <= 35:
        return 4
    return None


def paired(nt, t):
    return t < 2 * (nt // 2)


def colmap(nt, t):
    """(base column, stride) of fragment positions 0..15 of tile t"""
    return (32 * (t >> 1) + (t & 1), 2) if paired(nt, t) else (16 * t, 1)


def plan(nt, g, C):
    """-> list of 8 C roles in launch order (role index = member * 8 + wave); each a dict:
    tiles [(ti, tj)], tails [(t, gi)], a_set / b_set (tiles needed on each side), scale_a (bool), frags [tile ids]"""
    nw = 8 * C
    walk = []
    for b0 in range(0, nt, BAND):
        rows = list(range(b0, min(b0 + BAND, nt)))
        for tj in range(b0, nt):
            walk.extend((ti, tj) for ti in rows if ti <= tj)
    n = len(walk)
    runs = []
    for w in range(nw):
        lo, hi = (n * w) // nw, (n * (w + 1)) // nw
        runs.append({"tiles": walk[lo:hi], "tails": []})
    if g:
        load = [len(r["tiles"]) * 4 for r in runs]                 # in quarter-tiles (a 4x4x4 tail MFMA = 1)
        cap = (nt + 1 + nw - 1) // nw
        nrows_t = [0] * nw
        # a wave's accumulators must leave room for its ~86 VGPRs at two waves per SIMD: 8 per tile + 2 per tail entry <= TAIL_REG_CAP
        # (NT = 17 / 35 with three tail groups put 6 entries on 20-tile waves -- 172 registers -- before this cap existed and
        # were left to the panel kernel).  A tile row's g entries stay on one wave when one has room for all of them; otherwise
        # they are placed one by one (NT = 17, G = 3: 54 entries on seven 19-tile waves and one 20-tile wave).
        room = lambda q, k: 8 * len(runs[q]["tiles"]) + 2 * (len(runs[q]["tails"]) + k) <= TAIL_REG_CAP
        holds = lambda q, t: any(t in (ti, tj) for ti, tj in runs[q]["tiles"])
        for t in range(nt + 1):
            whole = [w for w in range(nw) if nrows_t[w] < cap and room(w, g)]
            if whole:
                cand = [w for w in whole if holds(w, t)] or whole
                w = min(cand, key=lambda q: load[q])
                runs[w]["tails"].extend((t, gi) for gi in range(g))
                load[w] += g
                nrows_t[w] += 1
                continue
            for gi in range(g):
                cand = [w for w in range(nw) if room(w, 1)] or list(range(nw))
                cand = [w for w in cand if holds(w, t)] or cand
                w = min(cand, key=lambda q: load[q])
                runs[w]["tails"].append((t, gi))
                load[w] += 1
    for r in runs:
        r["load"] = 4 * len(r["tiles"]) + len(r["tails"])
        a_set = sorted(set([ti for ti, _ in r["tiles"]] + [t for t, _ in r["tails"]]))
        b_set = sorted(set(tj for _, tj in r["tiles"]))
        r["a_set"], r["b_set"] = a_set, b_set
        # the weight multiplies the side with fewer operands (tail columns count on the B side)
        r["scale_a"] = len(a_set) <= len(b_set) + (g if r["tails"] else 0)
        r["frags"] = sorted(set(a_set) | set(b_set))
    # SIMD pairing: heaviest with lightest; pair k -> member k % C, SIMD k // C (waves s and s + 4 share SIMD s)
    order = sorted(range(nw), key=lambda w: -runs[w]["load"])
    roles = [None] * nw
    for k in range(nw // 2):
        member, simd = k % C, k // C
        roles[member * 8 + simd] = runs[order[k]]
        roles[member * 8 + simd + 4] = runs[order[nw - 1 - k]]
    # workgroups of a group share rows through L2 and must stay in lock step: a workgroup whose busiest SIMD is lighter than
    # the group's gets dummy 4x4x4 MFMAs (into a scratch accumulator pair) on that SIMD
    for r in roles:
        r["pad"] = 0
    simd_load = lambda m, s: roles[m * 8 + s]["load"] + roles[m * 8 + s + 4]["load"]
    top = max(simd_load(m, s) for m in range(C) for s in range(4))
    for m in range(C):
        s = max(range(4), key=lambda q: simd_load(m, q))
        roles[m * 8 + s]["pad"] = top - simd_load(m, s)
    return roles


def role_regs(r):
    return max(8 * len(r["tiles"]) + 2 * len(r["tails"]) + (2 if r["pad"] else 0), 8)


def loads_of(nt, r):
    """-> list of (kind, imm, [frag index, ...]): kind 'p2' = b128 of a pair, 'p1' = one tile of a pair, 'pl' = plain tile"""
    fr = r["frags"]
    out = []
    done = set()
    for t in fr:
        if t in done:
            continue
        if paired(nt, t):
            g = t >> 1
            both = (2 * g in fr) and (2 * g + 1 in fr)
            if both:
                out.append(("p2", 256 * g, [fr.index(2 * g), fr.index(2 * g + 1)]))
                done.update((2 * g, 2 * g + 1))
            else:
                out.append(("p1", 256 * g + 8 * (t & 1), [fr.index(t)]))
                done.add(t)
        else:
            out.append(("pl", 128 * t, [fr.index(t)]))
            done.add(t)
    return out


def emit(nt, g, C, out):
    roles = plan(nt, g, C)
    nw = 8 * C
    maxf = max(len(r["frags"]) for r in roles)
    maxs = max(len(r["a_set"]) if r["scale_a"] else len(r["b_set"]) for r in roles)
    # GP_LEVEL=1 (timing experiment): every role is given the LDS reads and multiplications of the heaviest one
    level = os.environ.get("GP_LEVEL") == "1"
    nmul = lambda r: (len(r["a_set"]) if r["scale_a"] else len(r["b_set"]) + (g if r["tails"] else 0))
    top_reads = max(len(loads_of(nt, r)) for r in roles)
    top_muls = max(nmul(r) for r in roles)
    for r in roles:
        r["xreads"] = top_reads - len(loads_of(nt, r)) if level else 0
        r["xmuls"] = top_muls - nmul(r) if level else 0
    maxx = max(max(r["xreads"], r["xmuls"]) for r in roles)
    maxf_real, maxs_real = maxf, maxs
    maxf += maxx
    maxs += maxx
    nreg = max(role_regs(r) for r in roles)
    assert nreg <= MAX_AGPR, (nt, g, C, nreg)
    ga = max(g, 1)
    ntc = nt + (1 if g else 0)
    npq = (32 * ((ntc + 1) // 2) * 8 + 1023) // 1024             # 1 KB DMA pieces per row: gram_plan.h plan_pitch()
    rpw = 2 if C == 1 else 1                                     # rows per wave and chunk: gram_plan.h plan_kc() / 8
    nseg = rpw * npq + 1
    out.append("// ---- NT = %d, G = %d, C = %d: tiles %s tails %s frags %s LDS reads %s" % (
        nt, g, C, [len(r["tiles"]) for r in roles], [len(r["tails"]) for r in roles], [len(r["frags"]) for r in roles],
        [len(loads_of(nt, r)) for r in roles]))
    out.append("template <> struct GPlan<%d, %d> {" % (nt, g))
    out.append("    static constexpr int C = %d, MAXF = %d, MAXS = %d, NREG = %d, NSEG = %d;" % (C, maxf, maxs, nreg, nseg))
    out.append("    template <int R, typename L2, typename L1, typename LP> static __device__ __forceinline__ void load(L2&& ld2, L1&& ld1, LP&& ldp, double (&f)[MAXF]) {")
    for w, r in enumerate(roles):
        stm = []
        for kind, imm, idx in loads_of(nt, r):
            if kind == "p2":
                stm.append("{ const dlsa_d2 q = ld2(%d); f[%d] = q.x; f[%d] = q.y; }" % (imm, idx[0], idx[1]))
            elif kind == "p1":
                stm.append("f[%d] = ld1(%d);" % (idx[0], imm))
            else:
                stm.append("f[%d] = ldp(%d);" % (idx[0], imm))
        first = loads_of(nt, r)[0]
        for k in range(max(r["xreads"], r["xmuls"])):
            stm.append("f[%d] = %s(%d);" % (maxf_real + k, "ld1" if first[0] != "pl" else "ldp", first[1] + 16 * (k + 1)))
        out.append("        %sif constexpr (R == %d) { %s }" % ("" if w == 0 else "else ", w, " ".join(stm)))
    out.append("    }")
    # the same loads in NSEG parts (part q holds the reads q, q + NSEG, ...): the second k-step of a chunk issues part q of
    # the NEXT k-step's fragment reads behind segment q of its MFMA block
    out.append("    template <int R, int Q, typename L2, typename L1, typename LP> static __device__ __forceinline__ void load_part(L2&& ld2, L1&& ld1, LP&& ldp, double (&f)[MAXF]) {")
    for w, r in enumerate(roles):
        out.append("        %sif constexpr (R == %d) {" % ("" if w == 0 else "else ", w))
        lds = loads_of(nt, r)
        for q in range(nseg):
            stm = []
            for kind, imm, idx in lds[q::nseg]:
                if kind == "p2":
                    stm.append("{ const dlsa_d2 q2 = ld2(%d); f[%d] = q2.x; f[%d] = q2.y; }" % (imm, idx[0], idx[1]))
                elif kind == "p1":
                    stm.append("f[%d] = ld1(%d);" % (idx[0], imm))
                else:
                    stm.append("f[%d] = ldp(%d);" % (idx[0], imm))
            out.append("            %sif constexpr (Q == %d) { %s }" % ("" if q == 0 else "else ", q, " ".join(stm)))
        out.append("        }")
    out.append("    }")
    out.append("    template <int R> static constexpr bool has_tails() { constexpr bool n[%d] = {%s}; return n[R]; }" % (
        nw, ", ".join("true" if r["tails"] else "false" for r in roles)))
    # scaled copies: s[k] = w * fragment of the k-th tile of the scaled side (bt too when that is the B side)
    out.append("    template <bool HASW, int R> static __device__ __forceinline__ void scale(double w, const double (&f)[MAXF], double (&s)[MAXS], double (&bt)[%d]) {" % ga)
    for w, r in enumerate(roles):
        side = r["a_set"] if r["scale_a"] else r["b_set"]
        stm = ["s[%d] = HASW ? f[%d] * w : f[%d];" % (k, r["frags"].index(t), r["frags"].index(t)) for k, t in enumerate(side)]
        if not r["scale_a"] and r["tails"]:
            stm += ["if (HASW) bt[%d] *= w;" % gi for gi in range(g)]
        for k in range(max(r["xreads"], r["xmuls"])):
            stm.append("s[%d] = f[%d] * w;" % (maxs_real + k, maxf_real + k) if k < r["xmuls"] else "s[%d] = f[%d];" % (maxs_real + k, maxf_real + k))
        out.append("        %sif constexpr (R == %d) { %s }" % ("" if w == 0 else "else ", w, " ".join(stm)))
    out.append("    }")
    # PART 0 = the whole k-step, 1 / 2 = its first / second half (the second wave of a SIMD meets the chunk barrier in the
    # middle of its MFMA block, so that its second half covers the first wave's post-barrier loads and vice versa)
    out.append("    template <int R, int PART> static __device__ __forceinline__ void mfma(const double (&f)[MAXF], const double (&s)[MAXS], const double (&bt)[%d]) {" % ga)
    for w, r in enumerate(roles):
        fr = r["frags"]
        side = r["a_set"] if r["scale_a"] else r["b_set"]
        ops, opidx = [], {}

        def op(expr):
            if expr not in opidx:
                opidx[expr] = len(ops)
                ops.append(expr)
            return opidx[expr]

        def a_op(t):
            return op("s[%d]" % side.index(t)) if r["scale_a"] else op("f[%d]" % fr.index(t))

        def b_op(t):
            return op("f[%d]" % fr.index(t)) if r["scale_a"] else op("s[%d]" % side.index(t))

        body = []
        for k, (ti, tj) in enumerate(r["tiles"]):
            body.append("v_mfma_f64_16x16x4_f64 a[%d:%d], %%%d, %%%d, a[%d:%d]" % (8 * k, 8 * k + 7, a_op(ti), b_op(tj), 8 * k, 8 * k + 7))
        base = 8 * len(r["tiles"])
        for k, (t, gi) in enumerate(r["tails"]):
            body.append("v_mfma_f64_4x4x4_4b_f64 a[%d:%d], %%%d, %%%d, a[%d:%d]" % (base + 2 * k, base + 2 * k + 1, a_op(t), op("bt[%d]" % gi), base + 2 * k, base + 2 * k + 1))
        top = base + 2 * len(r["tails"])
        for k in range(r["pad"]):        # lock-step padding: same operands as the first tile, scratch accumulator
            body.append("v_mfma_f64_4x4x4_4b_f64 a[%d:%d], %%0, %%1, a[%d:%d]" % (top, top + 1, top, top + 1))
        if r["pad"]:
            top += 2
        clob = ", ".join('"a%d"' % q for q in range(max(top, 1)))
        half = len(r["tiles"]) // 2
        for k in range(max(r["xreads"], r["xmuls"])):
            op("s[%d]" % (maxs_real + k))
        opstr = ", ".join('"v"(%s)' % e for e in ops)
        parts = [["s_nop 1"] + body, ["s_nop 1"] + body[:half], body[half:]]
        # PART 10 + k: segment k of NSEG consecutive near-equal segments (the second k-step of a chunk issues one DMA piece of
        # the chunk three ahead behind each segment)
        for k in range(nseg):
            seg = body[(len(body) * k) // nseg:(len(body) * (k + 1)) // nseg]
            parts.append((["s_nop 1"] if k == 0 else []) + seg)
        out.append("        %sif constexpr (R == %d) {" % ("" if w == 0 else "else ", w))
        for part, lines in enumerate(parts):
            pid = part if part < 3 else 10 + part - 3
            if not lines:
                lines = ["s_nop 0"]
            out.append('            %sif constexpr (PART == %d) asm volatile("%s" :: %s : %s);' % (
                "" if part == 0 else "else ", pid, "\\n\\t".join(lines), opstr, clob))
        out.append("        }")
    out.append("    }")
    out.append("    template <int R> static __device__ __forceinline__ void store(int lane, double* __restrict__ P, int PP) {")
    for w, r in enumerate(roles):
        stm = []
        for k, (ti, tj) in enumerate(r["tiles"]):
            rb, rs = colmap(nt, ti)
            cb, cs = colmap(nt, tj)
            fold = 1 if (ti != tj and paired(nt, ti) and paired(nt, tj) and (ti >> 1) == (tj >> 1)) else 0
            stm.append("gp_store_tile<%d, %d>(P, PP, lane, %d, %d, %d, %d);" % (k, fold, rb, rs, cb, cs))
        base = 8 * len(r["tiles"])
        for k, (t, gi) in enumerate(r["tails"]):
            rb, rs = colmap(nt, t)
            stm.append("gp_store_tail<%d>(P, PP, lane, %d, %d, %d);" % (base + 2 * k, rb, rs, 16 * nt + 4 * gi))
        out.append("        %sif constexpr (R == %d) { %s }" % ("" if w == 0 else "else ", w, " ".join(stm)))
    out.append("    }")
    out.append("};")


def emit_common(out):
    out.append("// GENERATED by tools/gen_gram_plan_asm.py common -- do not edit.")
    out.append("template <int NREG> __device__ __forceinline__ void gp_acc_zero() {")
    first = True
    for n in range(8, MAX_AGPR + 1, 2):
        body = "\\n\\t".join("v_accvgpr_write_b32 a%d, 0" % r for r in range(n))
        out.append('    %sif constexpr (NREG == %d) asm volatile("%s" ::: %s);' % ("" if first else "else ", n, body, ", ".join('"a%d"' % q for q in range(n))))
        first = False
    out.append("}")
    out.append("template <int K> __device__ __forceinline__ void gp_tile_read(double (&v)[4]) {")
    out.append("    int w0, w1, w2, w3, w4, w5, w6, w7;")
    for t in range(MAX_AGPR // 8):
        b = "\\n\\t".join("v_accvgpr_read_b32 %%%d, a%d" % (r, 8 * t + r) for r in range(8))
        out.append('    %sif constexpr (K == %d) asm volatile("%s" : "=v"(w0), "=v"(w1), "=v"(w2), "=v"(w3), "=v"(w4), "=v"(w5), "=v"(w6), "=v"(w7));'
                   % ("" if t == 0 else "else ", t, b))
    out.append("    v[0] = __hiloint2double(w1, w0); v[1] = __hiloint2double(w3, w2);")
    out.append("    v[2] = __hiloint2double(w5, w4); v[3] = __hiloint2double(w7, w6);")
    out.append("}")
    out.append("template <int R> __device__ __forceinline__ double gp_pair_read() {")
    out.append("    int lo, hi;")
    first = True
    for r in range(0, MAX_AGPR, 2):
        out.append('    %sif constexpr (R == %d) asm volatile("v_accvgpr_read_b32 %%0, a%d\\n\\tv_accvgpr_read_b32 %%1, a%d" : "=v"(lo), "=v"(hi));'
                   % ("" if first else "else ", r, r, r + 1))
        first = False
    out.append("    return __hiloint2double(hi, lo);")
    out.append("}")
    out.append("// tile K of the wave: C/D register q of lane l = C[4q + (l >> 4)][l & 15]; fragment position r of the row tile is")
    out.append("// column rb + rs r of H, position c of the column tile column cb + cs c.  FOLD (the two interleaved tiles of one")
    out.append("// 32-column group): elements below the diagonal are the transposes of upper elements no other tile computes.")
    out.append("template <int K, int FOLD> __device__ __forceinline__ void gp_store_tile(double* __restrict__ P, int PP, int lane, int rb, int rs, int cb, int cs) {")
    out.append("    double v[4];")
    out.append("    gp_tile_read<K>(v);")
    out.append("    const int col = cb + cs * (lane & 15);")
    out.append("#pragma unroll")
    out.append("    for (int q = 0; q < 4; ++q) {")
    out.append("        const int row = rb + rs * (4 * q + (lane >> 4));")
    out.append("        if (FOLD && row > col) P[(int64_t)col * PP + row] = v[q];")
    out.append("        else P[(int64_t)row * PP + col] = v[q];")
    out.append("    }")
    out.append("}")
    out.append("// tail accumulator at AGPR R = tile row t x tail columns c0 .. c0 + 3: lane l holds row position 4 b + i, column c0 + j,")
    out.append("// i = l >> 4, b = (l & 15) >> 2, j = l & 3")
    out.append("template <int R> __device__ __forceinline__ void gp_store_tail(double* __restrict__ P, int PP, int lane, int rb, int rs, int c0) {")
    out.append("    P[(int64_t)(rb + rs * (4 * ((lane & 15) >> 2) + (lane >> 4))) * PP + c0 + (lane & 3)] = gp_pair_read<R>();")
    out.append("}")
    out.append("template <int NT, int G> struct GPlan;")


def stats():
    for nt in range(8, 36):
        C = groups_for(nt)
        for g in range(MAXG + 1):
            roles = plan(nt, g, C)
            simd = {}
            for i, r in enumerate(roles):
                simd.setdefault((i // 8, i % 4), 0)
                simd[(i // 8, i % 4)] += r["load"] + r["pad"]
            wg = [max(v for (m, s), v in simd.items() if m == mm) for mm in range(C)]
            tot = sum(r["load"] for r in roles)
            reads = [len(loads_of(nt, r)) for r in roles]
            print("NT %2d G %d C %d: regs %3d  simd load max %3d mean %.1f (eff %.3f)  wg max %s  reads/wave mean %.1f max %d  frags max %d" % (
                nt, g, C, max(role_regs(r) for r in roles), max(simd.values()), tot / (4.0 * C), tot / (4.0 * C) / max(simd.values()),
                wg, sum(reads) / len(reads), max(reads), max(len(r["frags"]) for r in roles)))


def main():
    if len(sys.argv) >= 2 and sys.argv[1] == "common":
        out = []
        emit_common(out)
        print("\n".join(out))
    elif len(sys.argv) >= 5 and sys.argv[1] == "plans":
        C, lo, hi = int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
        out = ["// GENERATED by tools/gen_gram_plan_asm.py plans %d %d %d -- do not edit." % (C, lo, hi)]
        for nt in range(lo, hi + 1):
            assert groups_for(nt) == C, (nt, C)
            for g in range(MAXG + 1):
                emit(nt, g, C, out)
        print("\n".join(out))
    elif len(sys.argv) >= 2 and sys.argv[1] == "stats":
        stats()
    else:
        sys.exit(__doc__)


if __name__ == "__main__":
    main()
