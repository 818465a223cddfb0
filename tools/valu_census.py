#!/usr/bin/env python
"""Static instruction census of the C2 post kernel's phases (no GPU needed).

    python tools/valu_census.py [--out profiles/rN_c2_valu_census.txt]

Compiles tools/exp/census.hip (every phase of payne_post_kernel<12, true, true> as a kernel of its own, with
run_candidate's template arguments) for gfx950 and counts the instructions of each in the compiler's assembly:
vector ALU (of which packed fp32, fp64, conversions, compares / selects), LDS, global memory, scalar, s_nop, waits.
The phases' loops have compile-time trip counts, so the static count is what a wave executes -- except where a line
says otherwise (branches taken by part of the waves: the transform's radix-8 passes run on waves 0-3 only).
"""
import argparse
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tools", "exp", "census.hip")

CLASSES = ["valu", "pk_f32", "f64", "cvt", "cmp_sel", "lds", "vmem", "salu", "s_nop", "s_waitcnt", "s_barrier", "branch"]


def classify(op):
    c = []
    if op.startswith("v_"):
        c.append("valu")
        if op.startswith("v_pk_") and "f32" in op:
            c.append("pk_f32")
        if "f64" in op:
            c.append("f64")
        if op.startswith("v_cvt"):
            c.append("cvt")
        if op.startswith("v_cmp") or op.startswith("v_cndmask"):
            c.append("cmp_sel")
    elif op.startswith("ds_"):
        c.append("lds")
    elif op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        c.append("vmem")
    elif op == "s_nop":
        c.append("s_nop")
    elif op == "s_waitcnt":
        c.append("s_waitcnt")
    elif op == "s_barrier":
        c.append("s_barrier")
    elif op.startswith(("s_cbranch", "s_branch", "s_setpc", "s_swappc", "s_call")):
        c.append("branch")
    elif op.startswith("s_"):
        c.append("salu")
    return c


def census(asm_text):
    """{function: {class: count}} plus, per function, the counts between consecutive s_barrier instructions."""
    funcs, cur, segs = {}, None, None
    for line in asm_text.splitlines():
        m = re.match(r"^(_Z\w+):", line)
        if m:
            cur = m.group(1)
            funcs[cur] = {"total": dict.fromkeys(CLASSES, 0), "segments": [dict.fromkeys(CLASSES, 0)]}
            continue
        if cur is None:
            continue
        s = line.strip()
        if s.startswith(".Lfunc_end"):
            cur = None
            continue
        if not s or s.startswith((";", ".", "$")) or s.endswith(":"):
            continue
        op = s.split()[0]
        for c in classify(op):
            funcs[cur]["total"][c] += 1
            funcs[cur]["segments"][-1][c] += 1
        if op == "s_barrier":
            funcs[cur]["segments"].append(dict.fromkeys(CLASSES, 0))
    return funcs


def demangle(name):
    try:
        return subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip().split("(")[0]
    except OSError:
        return name


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=None)
    ap.add_argument("--asm", default=None, help="count an existing assembly file instead of compiling census.hip")
    a = ap.parse_args()
    if a.asm:
        text = open(a.asm).read()
    else:
        with tempfile.TemporaryDirectory() as d:
            out = os.path.join(d, "census.s")
            cmd = ["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-DNDEBUG", "--cuda-device-only", "-S",
                   "-I", os.path.join(ROOT, "include"), "-o", out, SRC]
            r = subprocess.run(cmd, capture_output=True, text=True)
            if r.returncode != 0:
                sys.exit(r.stderr)
            text = open(out).read()
    funcs = census(text)
    lines = []
    hdr = "%-34s" % "phase" + "".join("%10s" % c for c in CLASSES)
    lines.append(hdr)
    for name, f in funcs.items():
        short = demangle(name)
        if "vsini_sb_exact" in short:
            continue
        lines.append("%-34s" % short[-34:] + "".join("%10d" % f["total"][c] for c in CLASSES))
        if len(f["segments"]) > 2:
            for i, sgm in enumerate(f["segments"]):
                if sum(sgm.values()):
                    lines.append("%-34s" % ("   between barriers %d" % i) + "".join("%10d" % sgm[c] for c in CLASSES))
    txt = "\n".join(lines)
    print(txt)
    if a.out:
        with open(a.out, "w") as fh:
            fh.write(txt + "\n")


if __name__ == "__main__":
    main()
