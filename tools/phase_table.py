#!/usr/bin/env python3
"""Per-phase issue-floor table of a sweep kernel: instructions between consecutive phase stamps of the STAMPED build's own
assembly (every STAMP(i) leaves a `; HMCG_STAMP_MARK i` comment; the segment that ENDS at mark i is phase i) against the
ticks the same build measured for that phase (tools/stamps.py / tools/clock_stamps.py output).

Cost model (measured in round 4, profiles/r04/README.md): a wave alone on its SIMD issues ONE instruction of ANY kind per
~4.4 ticks (vector, scalar, LDS, memory, wait, nop alike), so the issue floor of a phase is 4.4 x its instruction count;
what exceeds it is waiting (barriers, s_waitcnt on LDS / memory, dependent-chain latency).

The count is STATIC and follows the listing: a segment is straight-line for the window's own waves in the unrolled
register-resident kernels, but it also contains the out-of-line rare paths (underflow, rejection retries) and, in the
parameter phase, the code of every wave role (wave 0 / shadow waves / helper waves) -- those rows are marked `roles`.
Loops inside a segment are listed with their body size (the table counts one trip).

    python tools/phase_table.py <stamped .s> '<demangled-name substring>' [stamps-output.txt] [wave-column]
"""
import collections
import re
import subprocess
import sys

NAMES = ["Ba wait", "param draws | shadow jobs", "Bb wait", "theta+ux+pdfs", "local product", "wave scan", "Bc wait",
         "prefix+replay+last", "Bd wait", "maps+compose", "map scan", "Be wait", "apply", "publish stats", "(shadow: outputs)",
         "(shadow: prep)", "(param: counts+row sums)", "(param: shapes)", "(param: gamma)", "unused"]
ROLES = {1, 14, 15, 16, 17, 18}


def cls(op):
    if op.startswith("v_") and "f64" in op:
        return "fp64"
    if op.startswith("v_"):
        return "valu"
    if op.startswith("s_waitcnt"):
        return "wait"
    if op.startswith("s_barrier"):
        return "barrier"
    if op.startswith("s_"):
        return "salu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "buffer_", "scratch_")):
        return "vmem"
    return "salu"


def main():
    path, want = sys.argv[1], sys.argv[2]
    ticks = {}
    if len(sys.argv) > 3:
        col = int(sys.argv[4]) if len(sys.argv) > 4 else 0
        for ln in open(sys.argv[3]):
            for i, nm in enumerate(NAMES):
                if ln.startswith(nm) or ln.strip().startswith(nm):
                    vals = ln[len(nm) + ln.index(nm[0]):].split() if False else ln.replace(nm, "").split()
                    try:
                        ticks[i] = float(vals[col])
                    except (IndexError, ValueError):
                        pass
    cur, body = None, []
    for ln in open(path):
        m = re.match(r"^(_Z\w+):", ln)
        if m:
            name = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
            cur = name if want in name else None
            continue
        if ln.startswith(".Lfunc_end"):
            cur = None
        if cur:
            body.append(ln.rstrip("\n"))
    # the sweep loop: from the first mark 0 to the last mark before it repeats
    segs, acc, labels, started = [], [], {}, False
    for ln in body:
        m = re.search(r"HMCG_STAMP_MARK (\d+)", ln)
        if m:
            i = int(m.group(1))
            if started:
                segs.append((i, acc))
            started, acc = True, []
            continue
        if not started:
            continue
        if re.match(r"^\.LBB\d+_\d+:", ln):
            acc.append(("label", ln.split(":")[0]))
        elif ln.startswith("\t") and ln.strip() and not ln.lstrip().startswith((".", ";")):
            acc.append(("inst", ln.split()))
    print("%-28s %6s | %5s %5s %5s %5s %5s %5s | %9s %9s %6s  %s" % ("phase (segment ending at its stamp)", "insts", "fp64", "valu", "salu", "lds", "vmem", "wait",
                                                                      "floor 4.4x", "measured", "ratio", "loops inside (label: body instructions)"))
    seen = set()
    for i, acc in segs:
        if i in seen:
            continue
        seen.add(i)
        insts = [t for k, t in acc if k == "inst"]
        c = collections.Counter(cls(t[0]) for t in insts)
        pos, labs = 0, {}
        loops = []
        for k, t in acc:
            if k == "label":
                labs[t] = pos
            else:
                if t[0].startswith(("s_cbranch", "s_branch")) and len(t) > 1 and t[1] in labs:
                    loops.append("%s: %d" % (t[1], pos + 1 - labs[t[1]]))
                pos += 1
        n = len(insts)
        fl = 4.4 * n
        ms = ticks.get(i)
        print("%-28s %6d | %5d %5d %5d %5d %5d %5d | %9.0f %9s %6s  %s%s" % (
            NAMES[i] if i < len(NAMES) else str(i), n, c["fp64"], c["valu"], c["salu"] + c["barrier"], c["lds"], c["vmem"], c["wait"], fl,
            "%.0f" % ms if ms is not None else "-", "%.2f" % (ms / fl) if ms and n else "-", "roles; " if i in ROLES else "", ", ".join(loops)))


if __name__ == "__main__":
    main()
