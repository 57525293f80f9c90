#!/usr/bin/env python3
"""HEAD-tree counter evidence for every dominant kernel of the path: one JSON per kernel under profiles/, each carrying the hash of
the sources it was taken from (tests/test_bench_cpu.py refuses a stale one).

    python3 bench/pmc_evidence.py [round-tag] [kernel-tag ...]          e.g.  python3 bench/pmc_evidence.py r04 fused_p100 wide_f32_p2000

Runs on the GPU box.  This process never touches the GPU: every counter pass is its own `rocprofv3 --kernel-trace --pmc ... -- python3
<driver>` child (kernel-trace only next to --pmc; separate passes for the SQ groups, GRBM, FETCH_SIZE and WRITE_SIZE as
MI355X_MICROARCH.md prescribes).  Derived figures per kernel:
  pipe_busy      = SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs / (GRBM_GUI_ACTIVE / 8 XCDs)
  hbm_bytes      = (2 * FETCH_SIZE + WRITE_SIZE) KB  (FETCH_SIZE counts 32-byte... the gfx950 correction of the guide: x2)
  valu_per_mfma  = (SQ_INSTS_VALU - SQ_INSTS_MFMA) / SQ_INSTS_MFMA   (SQ_INSTS_VALU counts the MFMAs too)
"""
import collections
import csv
import glob
import hashlib
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "dlsa_amd", "csrc")

# tag -> (driver script + arguments, substring that selects the kernel, the sources the kernel is built from (relative to dlsa_amd/csrc
# unless they contain a slash), algorithmic bytes and flops of ONE launch for the fractions)
KERNELS = {
    "cyclic_p500": dict(cmd=["bench/gram_quick.py", "10000000", "500", "2"], pat="gram_cyclic_kernel",
                        src=["gram.hip", "gram_cyclic.hip", "gram_cyclic_asm.inc", "common.h"], rows=10_000_000, p=500, elem=8, kind="gram"),
    "plan_p260": dict(cmd=["bench/gram_quick.py", "10000000", "260", "2"], pat="gram_plan_kernel",
                      src=["gram_plan.hip", "gram_plan_kernel.inc", "gram_plan_unit.hip", "gram_plan.h", "common.h", "tools/gen_gram_plan_asm.py"],
                      rows=10_000_000, p=260, elem=8, kind="gram"),
    "narrow_p100": dict(cmd=["bench/gram_quick.py", "10000000", "100", "2"], pat="gram_narrow_kernel",
                        src=["gram_narrow.hip", "gram_narrow_asm.inc", "common.h"], rows=10_000_000, p=100, elem=8, kind="gram"),
    "fused_p100": dict(cmd=["bench/fused_quick.py", "10000000", "100"], pat="irls_pass_narrow_kernel<false, true",
                       src=["irls_pass.hip", "gram_narrow_asm.inc", "logistic.h", "common.h"], rows=10_000_000, p=100, elem=8, kind="fused"),
    "wide_f32_p2000": dict(cmd=["bench/gram_time_f32.py", "6000000", "2000", "2"], pat="gram_wide_f32_kernel",
                           src=["gram_wide.hip", "common.h"], rows=6_000_000, p=2000, elem=4, kind="gram_now"),
    "logit_p50": dict(cmd=["bench/fused_quick.py", "10000000", "50"], pat="irls_pass_narrow_kernel<true, false",
                      src=["irls_pass.hip", "logistic.h", "common.h"], rows=10_000_000, p=50, elem=8, kind="logit"),
    "logit_p500": dict(cmd=["bench/logit_one.py", "10000000", "500"], pat="logit_kernel",
                       src=["logit.hip", "rowdot.h", "logistic.h", "common.h"], rows=10_000_000, p=500, elem=8, kind="logit"),
    "wide_syrk_p500": dict(cmd=["bench/wide_syrk_probe.py", "1000000", "500"], pat="wide_syrk_kernel",
                           src=["irls_wide.hip", "common.h"], rows=1_000_000, p=500, elem=2, kind="syrk"),
    "logit_img_p500": dict(cmd=["bench/wide_syrk_probe.py", "1000000", "500"], pat="logit_kernel<4, 4, true, false, true>",
                           src=["logit.hip", "rowdot.h", "logistic.h", "common.h"], rows=1_000_000, p=500, elem=8, kind="logit"),
    "onehot_c4": dict(cmd=["bench/onehot_one.py", "14000000", "14"], pat="oh_gram_kernel",
                      src=["onehot.hip", "common.h"], rows=1_000_000, p=260, elem=8, kind="onehot"),
}

PASSES = {
    "sq1": ["SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS",
            "SQ_VALU_MFMA_BUSY_CYCLES"],
    "sq2": ["SQ_INSTS_VALU", "SQ_INSTS_MFMA", "SQ_INSTS_LDS", "SQ_INSTS_SALU", "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE", "SQ_WAIT_INST_LDS",
            "SQ_INSTS_VMEM_RD"],
    "grbm": ["GRBM_GUI_ACTIVE"],
    "fetch": ["FETCH_SIZE"],
    "write": ["WRITE_SIZE"],
}


def sources_sha16(src):
    h = hashlib.sha256()
    for f in src:
        path = os.path.join(ROOT, f) if "/" in f else os.path.join(CSRC, f)
        with open(path, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def evidence_path(round_tag, tag):
    return os.path.join(ROOT, "profiles", "%s_pmc_%s.json" % (round_tag, tag))


ROUNDS = ("r06", "r05", "r04")           # newest first: a kernel untouched since an earlier round keeps that round's evidence


def latest_evidence_path(tag):
    for r in ROUNDS:
        if os.path.exists(evidence_path(r, tag)):
            return evidence_path(r, tag)
    return evidence_path(ROUNDS[0], tag)


def collect(round_tag, tag):
    k = KERNELS[tag]
    out = os.path.join(ROOT, "gpurun_out", "pmc_ev_%s" % tag)
    os.makedirs(out, exist_ok=True)
    env = dict(os.environ, TMPDIR="/tmp")
    for name, ctrs in PASSES.items():
        cmd = ["rocprofv3", "--kernel-trace", "--pmc"] + ctrs + ["-d", os.path.join(out, name), "-o", name, "--output-format", "csv", "--",
                                                                 "python3", os.path.join(ROOT, k["cmd"][0])] + k["cmd"][1:]
        with open(os.path.join(out, name + ".log"), "w") as log:
            rc = subprocess.call(cmd, cwd=ROOT, env=env, stdout=log, stderr=subprocess.STDOUT)
        if rc:
            print("[pmc_evidence] %s pass %s: rocprofv3 exit code %d (see %s)" % (tag, name, rc, os.path.join(out, name + ".log")), file=sys.stderr)
    res = {}
    for f in sorted(glob.glob(os.path.join(out, "*", "*_counter_collection.csv"))):
        agg = collections.defaultdict(lambda: collections.defaultdict(float))
        cnt = collections.Counter()
        for r in csv.DictReader(open(f)):
            name = r["Kernel_Name"].split("(")[0]
            agg[name][r["Counter_Name"]] += float(r["Counter_Value"])
            cnt[(name, r["Counter_Name"])] += 1
        for name, d in agg.items():
            if k["pat"] in name and "reduce" not in name:
                for c, v in d.items():
                    res.setdefault(name, {})[c] = v / cnt[(name, c)]
    dur = collections.defaultdict(list)
    for f in glob.glob(os.path.join(out, "sq1", "*_kernel_trace.csv")):
        for r in csv.DictReader(open(f)):
            name = r["Kernel_Name"].split("(")[0]
            if k["pat"] in name and "reduce" not in name:
                dur[name].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
    for name, d in res.items():
        if name in dur:
            d["duration_ms_under_pmc"] = sorted(dur[name])[len(dur[name]) // 2]
            d["launches"] = len(dur[name])
        if d.get("GRBM_GUI_ACTIVE") and d.get("SQ_VALU_MFMA_BUSY_CYCLES") is not None:
            d["pipe_busy"] = d["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024.0 / (d["GRBM_GUI_ACTIVE"] / 8.0)
        if "FETCH_SIZE" in d:
            d["hbm_bytes"] = (2.0 * d["FETCH_SIZE"] + d.get("WRITE_SIZE", 0.0)) * 1024.0
        if d.get("SQ_INSTS_MFMA"):
            d["valu_per_mfma"] = (d["SQ_INSTS_VALU"] - d["SQ_INSTS_MFMA"]) / d["SQ_INSTS_MFMA"]
    doc = {"tag": tag, "round": round_tag, "command": " ".join(k["cmd"]), "kernel_pattern": k["pat"], "sources": k["src"],
           "sources_sha16": sources_sha16(k["src"]), "rows": k["rows"], "p": k["p"], "kernels": res,
           "notes": "counters are per launch (mean over the launches of the run); SQ_* cycle counters tick once per four clocks"}
    with open(evidence_path(round_tag, tag), "w") as f:
        json.dump(doc, f, indent=1, sort_keys=True)
    for name, d in res.items():
        print("%s %s: %.3f ms under the counters, pipe busy %s, HBM bytes %s, VALU per MFMA %s" % (
            tag, name, d.get("duration_ms_under_pmc", float("nan")), "%.3f" % d["pipe_busy"] if "pipe_busy" in d else "-",
            "%.4g" % d["hbm_bytes"] if "hbm_bytes" in d else "-", "%.2f" % d["valu_per_mfma"] if "valu_per_mfma" in d else "-"), flush=True)
    if not res:
        print("[pmc_evidence] %s: no kernel matched %r" % (tag, k["pat"]), file=sys.stderr)


def main():
    round_tag = sys.argv[1] if len(sys.argv) > 1 else ROUNDS[0]
    tags = sys.argv[2:] or list(KERNELS)
    for t in tags:
        collect(round_tag, t)


if __name__ == "__main__":
    main()
