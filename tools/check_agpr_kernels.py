#!/usr/bin/env python3
"""Post-build check of the AGPR-resident Gram kernels in libdlsa_hip.so (gram_narrow / gram_plan / gram_cyclic).

Those kernels keep their loop-carried accumulators in NAMED AGPRs that the inline-asm MFMA blocks only list as clobbers:
nothing in the C++ model stops a future hipcc from using the same registers between two asm statements (for example as
VGPR spill slots, which on gfx950 are v_accvgpr_write / v_accvgpr_read pairs).  This script reads the code objects the
library actually ships and asserts, per kernel:
  * no scratch: .private_segment_fixed_size == 0, .vgpr_spill_count == 0, .sgpr_spill_count == 0;
  * .agpr_count == the accumulator count the generators planned (narrow_nreg / GPlan::NREG / 128 + 4 G);
  * VGPRs + AGPRs fit two waves per SIMD (<= 256) where the kernel is launched that way;
  * NO v_accvgpr_read / v_accvgpr_write inside any loop (a cycle of the kernel's control-flow graph, found with Tarjan's
    algorithm on the disassembly's basic blocks): the only accumulator moves are the zeroing prologue and the store
    epilogue, both straight-line code outside the row loop.
Run by tests/test_build_cpu.py and by `make check`.  Needs llvm-objcopy / llvm-readelf / llvm-objdump (ROCm's llvm/bin).
"""
import os
import re
import struct
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
LLVM = os.environ.get("DLSA_LLVM_BIN", "/opt/rocm/lib/llvm/bin")
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def tool(name):
    p = os.path.join(LLVM, name)
    return p if os.path.exists(p) else name


def code_objects(lib, tmp):
    """gfx950 code objects of every translation unit linked into the library (.hip_fatbin = concatenated bundles)."""
    fat = os.path.join(tmp, "fat.bin")
    subprocess.check_call([tool("llvm-objcopy"), "--dump-section", ".hip_fatbin=" + fat, lib])
    d = open(fat, "rb").read()
    out = []
    for m in re.finditer(MAGIC, d):
        p = m.start()
        (ne,) = struct.unpack_from("<Q", d, p + 24)
        o = p + 32
        for _ in range(ne):
            off, sz, ts = struct.unpack_from("<QQQ", d, o)
            o += 24
            triple = d[o:o + ts].decode()
            o += ts
            if "gfx950" in triple and sz:
                f = os.path.join(tmp, "co%d.elf" % len(out))
                open(f, "wb").write(d[p + off:p + off + sz])
                out.append(f)
    return out


def kernel_meta(co):
    """name -> dict of the integer fields of the kernel's metadata note"""
    txt = subprocess.check_output([tool("llvm-readelf"), "--notes", co], text=True)
    kernels, cur = {}, None
    for ln in txt.splitlines():
        m = re.match(r"\s*(-\s+)?\.(\w+):\s+(\S+)\s*$", ln)
        if not m:
            continue
        if m.group(1) and m.group(2) == "agpr_count":
            cur = {}
        if cur is None:
            continue
        key, val = m.group(2), m.group(3)
        if key == "name":
            kernels[val] = cur
        elif re.fullmatch(r"-?\d+", val):
            cur[key] = int(val)
    return kernels


def _cyclic_blocks(ins):
    """Instructions -> set of addresses that lie on a cycle of the control-flow graph (a real loop, whatever the block
    layout: the role copies of the plan kernel are laid out so that plain backward JUMPS span other roles' prologues)."""
    addrs = [a for a, _, _ in ins]
    index = {a: i for i, a in enumerate(addrs)}
    leaders = {0}
    for i, (a, op, tgt) in enumerate(ins):
        if op.startswith("s_cbranch") or op in ("s_branch", "s_endpgm", "s_setpc_b64"):
            if i + 1 < len(ins):
                leaders.add(i + 1)
            if tgt is not None and tgt in index:
                leaders.add(index[tgt])
    starts = sorted(leaders)
    block_of = {}
    for b, s0 in enumerate(starts):
        e = starts[b + 1] if b + 1 < len(starts) else len(ins)
        for i in range(s0, e):
            block_of[i] = b
    succ = [[] for _ in starts]
    for b, s0 in enumerate(starts):
        e = (starts[b + 1] if b + 1 < len(starts) else len(ins)) - 1
        a, op, tgt = ins[e]
        if (op.startswith("s_cbranch") or op == "s_branch") and tgt is not None and tgt in index:
            succ[b].append(block_of[index[tgt]])
        if op not in ("s_branch", "s_endpgm", "s_setpc_b64") and e + 1 < len(ins):
            succ[b].append(block_of[e + 1])
    # iterative Tarjan: blocks in a strongly connected component of size > 1 (or with a self edge) are in a loop
    n = len(starts)
    idx, low, on, stack, cyc, counter = [-1] * n, [0] * n, [False] * n, [], set(), [0]
    for root in range(n):
        if idx[root] != -1:
            continue
        work = [(root, 0)]
        while work:
            v, pi = work.pop()
            if pi == 0:
                idx[v] = low[v] = counter[0]; counter[0] += 1
                stack.append(v); on[v] = True
            recurse = False
            for k in range(pi, len(succ[v])):
                w = succ[v][k]
                if idx[w] == -1:
                    work.append((v, k + 1)); work.append((w, 0)); recurse = True
                    break
                if on[w]:
                    low[v] = min(low[v], idx[w])
            if recurse:
                continue
            if low[v] == idx[v]:
                comp = []
                while True:
                    w = stack.pop(); on[w] = False; comp.append(w)
                    if w == v:
                        break
                if len(comp) > 1 or v in succ[v]:
                    cyc.update(comp)
            if work:
                u = work[-1][0]
                low[u] = min(low[u], low[v])
    return {addrs[i] for i in range(len(ins)) if block_of[i] in cyc}, len(cyc)


def loops_with_acc_moves(co, want):
    """name -> (v_accvgpr_* instructions inside a loop, blocks on loops, v_accvgpr_* in the kernel, MFMAs inside loops),
    for the kernels in `want`"""
    txt = subprocess.check_output([tool("llvm-objdump"), "-d", co], text=True)
    res, name, start, ins = {}, None, 0, []

    def finish():
        if name is None or name not in want:
            return
        inloop, nblocks = _cyclic_blocks(ins)
        bad = sum(1 for a, op, _ in ins if op.startswith("v_accvgpr_") and a in inloop)
        mf = sum(1 for a, op, _ in ins if op.startswith("v_mfma") and a in inloop)
        res[name] = (bad, nblocks, sum(1 for _, op, _ in ins if op.startswith("v_accvgpr_")), mf)

    for ln in txt.splitlines():
        m = re.match(r"^([0-9a-f]+) <(\S+)>:", ln)
        if m:
            finish()
            name, start, ins = m.group(2), int(m.group(1), 16), []
            continue
        m = re.match(r"^\s+(\S+).*//\s*([0-9A-Fa-f]+):", ln)
        if not m or name is None:
            continue
        op, addr = m.group(1), int(m.group(2), 16)
        tgt = None
        if op.startswith("s_cbranch") or op == "s_branch":
            t = re.search(r"<%s\+0x([0-9a-f]+)>" % re.escape(name), ln)
            tgt = start + int(t.group(1), 16) if t else (start if ("<%s>" % name) in ln else None)
        ins.append((addr, op, tgt))
    finish()
    return res


def expected_agprs(name):
    """(planned accumulator registers, max VGPR + AGPR) of an AGPR-resident kernel, or None for other kernels"""
    m = re.search(r"gram_narrow_kernelILb[01]ELi(\d+)ELi(\d+)EE", name)
    if m:
        nt, g = int(m.group(1)), int(m.group(2))
        full = 8 * (nt * (nt + 1) // 2) + 2 * (nt + 1) * g          # gram_narrow.hip: narrow_nreg_full (decides the workgroups per CU)
        d4 = 8 * (nt * (nt - 1) // 2) + 2 * (3 * nt + (nt + 1) * g)        # narrow_nreg under -DDLSA_NARROW_DIAG4=1 (round 6 variant: diagonal tiles are three pairs)
        nreg = d4 if os.environ.get("DLSA_CHECK_NARROW_DIAG4") else full
        return nreg, (256 if full <= 184 else 512)                 # narrow_wgs_per_cu: two workgroups of 4 waves per CU
    m = re.search(r"irls_pass_narrow_kernelILb[01]ELb1ELi(\d+)ELi(\d+)ELb[01]ELb[01]EE", name)
    if m:                                                          # irls_pass.hip: one wave per SIMD
        nt, g = int(m.group(1)), int(m.group(2))
        return 8 * (nt * (nt + 1) // 2) + 2 * (nt + 1) * g, 512
    m = re.search(r"gram_cyclic_kernelILb[01]ELi(\d+)EE", name)
    if m:
        return 128 + 4 * int(m.group(1)), 256
    m = re.search(r"gram_plan_kernelILb[01]ELi(\d+)ELi(\d+)EE", name)
    if m:
        import gen_gram_plan_asm as gp
        nt, g = int(m.group(1)), int(m.group(2))
        return max(gp.role_regs(r) for r in gp.plan(nt, g, gp.groups_for(nt))), 256
    return None


def check(lib=None, verbose=False):
    lib = lib or os.path.join(ROOT, "dlsa_amd", "libdlsa_hip.so")
    problems, seen = [], 0
    with tempfile.TemporaryDirectory() as tmp:
        for co in code_objects(lib, tmp):
            meta = kernel_meta(co)
            want = {k for k in meta if expected_agprs(k) is not None}
            if not want:
                continue
            moves = loops_with_acc_moves(co, want)
            for k in sorted(want):
                seen += 1
                md, (nreg, cap) = meta[k], expected_agprs(k)
                err = []
                for f in ("private_segment_fixed_size", "vgpr_spill_count", "sgpr_spill_count"):
                    if md.get(f, 0) != 0:
                        err.append("%s = %d" % (f, md[f]))
                # the allocation is rounded up to the next multiple of 4 registers at most (v_accvgpr pairs / alignment)
                if not (nreg <= md.get("agpr_count", -1) <= nreg + 3):
                    err.append("agpr_count %s, planned %d" % (md.get("agpr_count"), nreg))
                if md.get("vgpr_count", 0) > cap:
                    err.append("vgpr_count (VGPR + AGPR) %d > %d" % (md["vgpr_count"], cap))
                bad, nloops, total, mfma = moves.get(k, (None, 0, 0, 0))
                if bad is None:
                    err.append("not found in the disassembly")
                elif bad:
                    err.append("%d v_accvgpr moves inside a loop (%d loop blocks, %d moves in the kernel)" % (bad, nloops, total))
                elif nloops == 0 or mfma == 0:
                    err.append("no loop with MFMAs found (disassembly format changed?)")
                if verbose:
                    print("%-70s agpr %3d vgpr+agpr %3d loop blocks %d (MFMAs in loops %d) acc moves %d%s" % (
                        k[:70], md.get("agpr_count", -1), md.get("vgpr_count", -1), nloops, mfma, total, "  <-- " + "; ".join(err) if err else ""))
                if err:
                    problems.append((k, err))
    return seen, problems


if __name__ == "__main__":
    n, bad = check(sys.argv[1] if len(sys.argv) > 1 else None, verbose=True)
    print("%d AGPR-resident kernels checked, %d with problems" % (n, len(bad)))
    for k, err in bad:
        print("  %s: %s" % (k, "; ".join(err)))
    sys.exit(1 if bad or n == 0 else 0)
