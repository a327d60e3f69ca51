#!/usr/bin/env python3
"""Per-loop instruction table of one kernel, read from the compiler's own assembly (hipcc -save-temps=obj keeps it).

A loop is a backward branch: `s_cbranch_* .LBBf_n` / `s_branch .LBBf_n` to a label defined earlier in the function; its
body is everything from the label to the branch (nested loops are listed by themselves and counted in their parents).
For every loop: instructions by class -- fp64 arithmetic, other VALU, v_mov / v_accvgpr / lane moves (data movement that
does no arithmetic), v_cndmask, SALU, LDS, VMEM, waits.  The straight-line remainder of the function is the last row.

    python tools/isa_loops.py <file.s> '<demangled-name substring>' [--min N]
"""
import collections
import re
import subprocess
import sys

CLASSES = ("fp64", "valu", "vmov", "accvgpr", "lane", "cndmask", "salu", "lds", "vmem", "wait", "nop")


def cls(op):
    if op.startswith("v_accvgpr"):
        return "accvgpr"
    if op.startswith(("v_readlane", "v_writelane", "v_readfirstlane")):
        return "lane"
    if op.startswith("v_mov"):
        return "vmov"
    if op.startswith("v_cndmask"):
        return "cndmask"
    if op.startswith("v_") and "f64" in op:
        return "fp64"
    if op.startswith("v_"):
        return "valu"
    if op.startswith("s_waitcnt"):
        return "wait"
    if op.startswith("s_nop"):
        return "nop"
    if op.startswith("s_"):
        return "salu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "buffer_", "scratch_", "flat_")):
        return "vmem"
    return "salu"


def functions(path):
    cur, out = None, {}
    for ln in open(path):
        m = re.match(r"^(_Z\w+):", ln)
        if m:
            cur = m.group(1)
            out[cur] = []
            continue
        if ln.startswith(".Lfunc_end"):
            cur = None
        if cur is not None:
            out[cur].append(ln.rstrip("\n"))
    return out


def main():
    path, want = sys.argv[1], sys.argv[2]
    minn = int(sys.argv[sys.argv.index("--min") + 1]) if "--min" in sys.argv else 40
    for name, lines in functions(path).items():
        dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
        if want not in dem:
            continue
        labels, insts = {}, []          # label -> index into insts
        for ln in lines:
            m = re.match(r"^(\.LBB\d+_\d+):", ln)
            if m:
                labels[m.group(1)] = len(insts)
                continue
            if ln.startswith("\t") and ln.strip() and not ln.lstrip().startswith((".", ";")):
                insts.append(ln.split())
        loops = []
        for i, tok in enumerate(insts):
            if tok[0].startswith(("s_cbranch", "s_branch")) and len(tok) > 1 and tok[1] in labels and labels[tok[1]] <= i:
                loops.append((labels[tok[1]], i, tok[1]))
        print(dem)
        print("%-12s %6s %6s | " % ("loop", "insts", "VALU") + " ".join("%7s" % c for c in CLASSES))
        inloop = [False] * len(insts)
        for a, b, lab in sorted(loops):
            c = collections.Counter(cls(t[0]) for t in insts[a:b + 1])
            for j in range(a, b + 1):
                inloop[j] = True
            n = b + 1 - a
            if n < minn:
                continue
            valu = sum(c[k] for k in ("fp64", "valu", "vmov", "accvgpr", "lane", "cndmask"))
            print("%-12s %6d %6d | " % (lab, n, valu) + " ".join("%7d" % c[k] for k in CLASSES))
        c = collections.Counter(cls(t[0]) for j, t in enumerate(insts) if not inloop[j])
        valu = sum(c[k] for k in ("fp64", "valu", "vmov", "accvgpr", "lane", "cndmask"))
        print("%-12s %6d %6d | " % ("straight", sum(c.values()), valu) + " ".join("%7d" % c[k] for k in CLASSES))
        c = collections.Counter(cls(t[0]) for t in insts)
        valu = sum(c[k] for k in ("fp64", "valu", "vmov", "accvgpr", "lane", "cndmask"))
        print("%-12s %6d %6d | " % ("function", len(insts), valu) + " ".join("%7d" % c[k] for k in CLASSES))


if __name__ == "__main__":
    main()
