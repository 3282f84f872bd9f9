#!/usr/bin/env python3
"""isa_report.py — per-kernel resource table and hot-loop instruction histogram from the gfx950 assembly.

    python tools/isa_report.py [--build] [--asm PATH] [--kernel SUBSTR] [--loop]

--build compiles therldaisyworld_amd/csrc/dw_api.hip with --save-temps into /tmp/dw_isa (the flags of
therldaisyworld_amd/build.py) and reads the resulting .s; otherwise --asm names an existing one.
For every kernel whose (demangled) name contains SUBSTR: VGPRs, AGPRs, SGPRs, spills, scratch bytes,
LDS bytes, occupancy, code length; with --loop also the instruction-class histogram of the largest
loop body (the block between the last backward branch target and its branch) and the lines that touch
scratch memory.  Runs in the build container (no GPU).
"""
from __future__ import annotations

import argparse
import collections
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = "/tmp/dw_isa"


def build(extra):
    sys.path.insert(0, ROOT)
    from therldaisyworld_amd import build as B
    os.makedirs(OUT, exist_ok=True)
    cmd = ["/opt/rocm/bin/hipcc", *B.FLAGS, *extra, "--save-temps", "-o", os.path.join(OUT, "lib.so"),
           os.path.join(B.CSRC, "dw_api.hip")]
    subprocess.check_call(cmd, cwd=OUT)
    return os.path.join(OUT, "dw_api-hip-amdgcn-amd-amdhsa-gfx950.s")


def demangle(names):
    # binutils' c++filt does not know the _Float16 mangling (DF16_): substitute a known type first
    p = subprocess.run(["c++filt"], input="\n".join(n.replace("DF16_", "Dh") for n in names), text=True, capture_output=True)
    return [d.replace("__fp16", "_Float16").replace("half", "_Float16") for d in p.stdout.split("\n")]


def classify(op):
    if op.startswith("v_pk_"):
        return "v_pk"
    if op.startswith(("v_sqrt", "v_rcp", "v_rsq", "v_exp", "v_log", "v_sin", "v_cos")):
        return "v_trans"
    if "dpp" in op:
        return "v_dpp"
    if op.startswith("v_cvt"):
        return "v_cvt"
    if op.startswith("v_mov") or op.startswith("v_accvgpr"):
        return "v_mov"
    if op.startswith("v_cmp") or op.startswith("v_cndmask"):
        return "v_cmp/cndmask"
    if op.startswith("v_readlane") or op.startswith("v_writelane") or op.startswith("v_readfirstlane"):
        return "v_lane"
    if op.startswith("v_"):
        return "v_other"
    if op.startswith("s_waitcnt"):
        return "s_waitcnt"
    if op.startswith("s_nop"):
        return "s_nop"
    if op.startswith("s_"):
        return "s_other"
    if op.startswith(("global_", "buffer_", "flat_")):
        return "vmem"
    if op.startswith("scratch_"):
        return "scratch"
    if op.startswith("ds_"):
        return "lds"
    return "other"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--build", action="store_true")
    ap.add_argument("--asm", default=os.path.join(OUT, "dw_api-hip-amdgcn-amd-amdhsa-gfx950.s"))
    ap.add_argument("--kernel", default="")
    ap.add_argument("--loop", action="store_true")
    ap.add_argument("--dump", default="", help="write the hot loop of the (single) selected kernel to this file")
    ap.add_argument("--flag", action="append", default=[], help="extra compiler flag for --build (repeatable)")
    a = ap.parse_args()
    path = build(a.flag) if a.build else a.asm
    text = open(path).read()
    # split into functions: "name:" ... ".end_amdhsa_kernel" blocks carry the metadata
    kern = re.findall(r"\.amdhsa_kernel (\S+)\n(.*?)\.end_amdhsa_kernel", text, re.S)
    names = [k for k, _ in kern]
    dem = dict(zip(names, demangle(names)))
    rows = []
    for name, meta in kern:
        d = dem[name]
        if a.kernel and a.kernel not in d:
            continue
        def g(key, default="0"):
            m = re.search(r"\.amdhsa_" + key + r" (\S+)", meta)
            return m.group(1) if m else default
        # the human-readable comment block after the code has the real numbers
        m = re.search(re.escape(name) + r":.*?; Kernel info:(.*?)(?=\n\t\.(?:text|section)|\Z)", text, re.S)
        info = m.group(1) if m else ""
        def gi(key):
            mm = re.search(r"; " + key + r"\s*[:=] (\d+)", info)
            return int(mm.group(1)) if mm else -1
        rows.append((d, gi("NumVgprs"), gi("NumAgprs"), gi("TotalNumSgprs"), gi("ScratchSize"), gi("Occupancy"),
                     gi("LDSByteSize"), gi("codeLenInByte"), name))
    rows.sort()
    print(f"{'VGPR':>5} {'AGPR':>5} {'SGPR':>5} {'scr B':>6} {'occ':>4} {'LDS':>7} {'code':>7}  kernel")
    for d, v, ag, s, sc, oc, lds, code, name in rows:
        short = re.sub(r"\(.*", "", d)
        print(f"{v:5d} {ag:5d} {s:5d} {sc:6d} {oc:4d} {lds:7d} {code:7d}  {short}")
        if a.loop:
            m = re.search(r"\n" + re.escape(name) + r":[^\n]*\n(.*?)\n\.Lfunc_end", text, re.S)
            if not m:
                continue
            lines = m.group(1).split("\n")
            labels = {}
            for i, ln in enumerate(lines):
                mm = re.match(r"(\.LBB\S+):", ln)
                if mm:
                    labels[mm.group(1)] = i
            best, best_pk = None, -1
            for i, ln in enumerate(lines):
                mm = re.match(r"\ts_cbranch_\S+ (\.LBB\S+)", ln) or re.match(r"\ts_branch (\.LBB\S+)", ln)
                if mm and mm.group(1) in labels and labels[mm.group(1)] < i:
                    span = (labels[mm.group(1)], i)
                    npk = sum(1 for x in lines[span[0]:span[1] + 1] if x.startswith("\tv_pk_"))   # the map's loop
                    if npk > best_pk:
                        best, best_pk = span, npk
            if best:
                hist = collections.Counter()
                n = 0
                for ln in lines[best[0]:best[1] + 1]:
                    mm = re.match(r"\t([a-z_0-9]+)", ln)
                    if mm and not ln.startswith("\t."):
                        hist[classify(mm.group(1))] += 1
                        n += 1
                valu = sum(c for k, c in hist.items() if k.startswith("v_"))
                if a.dump:
                    open(a.dump, "w").write("\n".join(lines[best[0]:best[1] + 1]) + "\n")
                print(f"      hot loop (most packed float32): {n} instructions, {valu} VALU: " +
                      ", ".join(f"{k} {c}" for k, c in sorted(hist.items(), key=lambda kv: -kv[1])))
            scr = [ln.strip() for ln in lines if re.match(r"\tscratch_", ln)]
            if scr:
                inloop = sum(1 for i, ln in enumerate(lines) if re.match(r"\tscratch_", ln) and best and best[0] <= i <= best[1])
                print(f"      scratch instructions: {len(scr)} ({inloop} inside the hot loop (most packed float32))")


if __name__ == "__main__":
    main()
