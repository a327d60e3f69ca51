#!/usr/bin/env python3
"""ISA lint for the gfx950 kernels of libhmcgibbs (run on the compiler's own assembly, csrc/obj/*.s).

What it looks for -- the miscompile behind round 1's HSA_STATUS_ERROR_MEMORY_APERTURE_VIOLATION:
the backend placed a VGPR -> AGPR live-range copy (`v_accvgpr_write_b32 a0/a1`, the per-lane output pointer
`out_base`) into the *prologue* of a join block, among the SGPR-spill `v_writelane`s and AHEAD of the
`s_or_b64 exec, exec, s[..]` that re-activates the lanes of the other branch.  Those lanes kept whatever the
AGPRs held before, later read it back as a pointer and stored through it.  Whether it happens depends on
register allocation, i.e. on unrelated source changes, so every build is checked:

  rule 1  in any basic block, no register-allocator copy (v_accvgpr_*, scratch_* spill/reload, plain v_mov vA, vB) may
          sit between the block start and the exec-widening instruction that opens it (s_or_b64 exec, exec, sN /
          s_or_saveexec_b64) when only prologue-class instructions (SALU, v_writelane/v_readlane, s_nop,
          s_waitcnt) and other such copies precede it;
  rule 2  no flat_* memory instruction (every access is global_*, scratch_* or ds_*) and no device-side function
          call (an un-inlined lambda captures by reference: flat accesses, a stack, s_swappc).
Kernels that spill to scratch are listed as warnings (a performance matter, not a correctness one).

Every rule-1 hit is classified by a walk over the divergent region it closes: the exec restore names the SGPR pair that
holds the saved mask; the walk goes back to the instruction that saved it (s_and_saveexec_b64 / s_mov_b64 sN, exec / s_xor
...) and asks whether the copied VALUE (the copy's source registers) was written inside that region.  If it was, the copy is
"phi-like": the value exists only in the lanes that are active there, and copying it under their mask is what a PHI needs.
If it was not -- the value is live-in from before the branch, every lane holds one -- the copy (or spill) under the narrow
mask loses the other lanes' values: "live-in", the round-1 fault.  When the saving instruction cannot be found (the mask came
back from an SGPR spill lane) the hit counts as live-in.  Both classes fail the build; --allow-phi lets phi-like ones pass
(diagnostics: to see whether a refused build is refused on the real pattern).

Exit status 1 when a rule fails.  Also prints, per kernel, VGPR/AGPR/spill/scratch/LDS figures (--table).
"""
import glob
import os
import re
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
OBJ = os.path.join(os.path.dirname(HERE), "hmc.jl_amd", "csrc", "obj")

# register-allocator traffic: AGPR copies, scratch spills/reloads, and plain VGPR->VGPR moves (live-range splits / late
# copies).  A join block's first real instruction is the exec restore, so such an instruction ahead of it -- with nothing
# but prologue-class instructions around it -- was put there under the wrong exec mask.  Plain moves are included on
# evidence: round 2 saw `v_mov_b32 v208, v137` in this position in gibbs<3,8,256,SIG,p2>, and that build wrote the
# last draw of every output lane from lane 0's role (tests/test_gpu_variants.py caught it).  A build that trips the rule
# on a harmless per-edge PHI copy has to be perturbed too -- the gate is deliberately conservative.
VEC_SPILL = re.compile(r"^\s*(v_accvgpr_(write|read|mov)_b32\s|scratch_(store|load)\w*\s|buffer_(store|load)\w*\s|"
                       r"v_mov_b32_e32\s+v\d+,\s*v\d+\s*$|v_mov_b64_e32\s+v\[\d+:\d+\],\s*v\[\d+:\d+\]\s*$)")
PROLOGUE_OK = re.compile(r"^\s*(s_|v_writelane_b32|v_readlane_b32)")
# exec-WIDENING instructions that end a divergent region (SI_END_CF -> s_or_b64 exec, exec, sN) or flip to the else
# side (s_or_saveexec_b64).  `s_and_b64 ..; s_mov_b64 exec, ..` NARROWS exec (a region begins): loads ahead of it are fine.
EXEC_WIDEN = re.compile(r"^\s*(s_or_b64\s+exec,\s*exec,|s_or_saveexec_b64)")
EXEC_OTHER = re.compile(r"^\s*s_\w+\s+exec,")
BLOCK_START = re.compile(r"^(\.LBB\d+_\d+:|; %bb\.\d+:|[_A-Za-z][\w$.]*:)")
KERNEL_START = re.compile(r"^(_Z\w+):\s")


def _regs(tok):
    """register numbers named by one operand token: v12 -> {12}, v[4:7] -> {4..7}, a3 -> {} (only VGPRs matter here)"""
    m = re.match(r"^-?\|?v(\d+)\|?$", tok)
    if m:
        return {int(m.group(1))}
    m = re.match(r"^-?\|?v\[(\d+):(\d+)\]\|?$", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    return set()


def _operands(s):
    body = s.split(None, 1)
    return [t.strip() for t in body[1].split(",")] if len(body) > 1 else []


def _copy_sources(s):
    """VGPRs whose VALUE a register-allocator copy / spill moves"""
    ops = _operands(s)
    if s.startswith(("scratch_store", "buffer_store")):
        return set().union(*[_regs(t) for t in ops[:2]]) if ops else set()
    if s.startswith(("v_mov_b32", "v_mov_b64", "v_accvgpr_write")):
        return _regs(ops[1]) if len(ops) > 1 else set()
    return set()          # reloads (scratch_load, v_accvgpr_read): the value comes from memory / an AGPR -- treated as live-in


def classify(lines, copy_idx, widen_idx):
    """'phi-like' if every source register of the copy at lines[copy_idx] is written inside the divergent region that the
    exec restore at lines[widen_idx] closes, else 'live-in'."""
    m = re.search(r"exec,\s*exec,\s*(s\[\d+:\d+\])", lines[widen_idx]) or re.search(r"s_or_saveexec_b64\s+\S+,\s*(s\[\d+:\d+\])", lines[widen_idx])
    if not m:
        return "live-in"
    mask = re.escape(m.group(1))
    src = _copy_sources(lines[copy_idx].split(";")[0].strip())
    if not src:
        return "live-in"
    written = set()
    for k in range(copy_idx - 1, max(copy_idx - 20000, -1), -1):
        t = lines[k].split(";")[0].strip()
        if not t or t.startswith("."):
            if KERNEL_START.match(lines[k]):
                break
            continue
        if re.match(r"^(s_and_saveexec_b64|s_or_saveexec_b64|s_xor_saveexec_b64)\s+" + mask, t) or re.match(r"^s_mov_b64\s+" + mask + r",\s*exec", t) \
                or re.match(r"^s_(and|xor|andn2)_b64\s+" + mask + r",\s*exec", t):
            return "phi-like" if src <= written else "live-in"
        if re.match(r"^(v_readlane_b32|s_load|s_mov_b64|s_mov_b32)\s+s", t) and re.search(mask.replace("\\[", "").split(":")[0] + r"\b", t.split(",")[0]):
            return "live-in"          # the mask came back from somewhere else: region start unknown
        if t.startswith(("v_", "ds_read", "ds_bpermute", "ds_swizzle", "global_load", "scratch_load", "buffer_load")) and not t.startswith(("v_cmp", "v_writelane")):
            ops = _operands(t)
            if ops:
                written |= _regs(ops[0])
    return "live-in"


def lint_file(path, allow_phi=False):
    problems, table = [], []
    kernel = None
    in_prologue = False
    last_getpc = -100
    pending = []            # vector spill instructions seen in the current block prologue
    with open(path) as f:
        lines = f.readlines()
    meta = {}
    for n, raw in enumerate(lines, 1):
        line = raw.split(";")[0].rstrip() if not raw.lstrip().startswith(";") else raw.rstrip()
        m = KERNEL_START.match(raw)
        if m:
            kernel = m.group(1)
        if BLOCK_START.match(raw.strip()) or BLOCK_START.match(raw):
            in_prologue, pending = True, []
            continue
        s = line.strip()
        if not s or s.startswith(";") or s.startswith("."):
            mm = re.match(r"\s*\.(amdhsa_\w+|\w+):?\s+(\S+)", raw)
            continue
        if re.match(r"^\s*flat_", line):
            problems.append((path, n, kernel, "flat_* memory instruction", s))
        if re.match(r"^\s*s_getpc_b64", line):
            last_getpc = n
        if re.match(r"^\s*(s_swappc_b64|s_call_b64)", line) or (re.match(r"^\s*s_setpc_b64", line) and n - last_getpc > 6):
            # (s_getpc + s_add + s_setpc within a few lines is branch relaxation -- a long jump in a very large kernel)
            problems.append((path, n, kernel, "function call / return (a device lambda was not inlined)", s))
        if in_prologue:
            if EXEC_WIDEN.match(line):
                if pending:
                    for pn, ps in pending:
                        kind = classify(lines, pn - 1, n - 1)
                        if kind == "phi-like" and allow_phi:
                            continue
                        problems.append((path, pn, kernel, "vector spill/copy ahead of the exec restore at line %d [%s]" % (n, kind), ps))
                in_prologue = False      # one exec restore per prologue is what the backend emits; stop here
            elif EXEC_OTHER.match(line):
                in_prologue = False      # exec narrowed / rewritten: not the join pattern
            elif VEC_SPILL.match(line):
                pending.append((n, s))
            elif PROLOGUE_OK.match(line):
                pass
            else:
                in_prologue = False
    # resource table from the metadata notes (one YAML-ish block per kernel)
    cur = {}
    for raw in lines:
        m = re.match(r"\s+- \.agpr_count:\s+(\d+)", raw) or re.match(r"\s+\.agpr_count:\s+(\d+)", raw)
        if m:
            cur["agpr"] = int(m.group(1))
        for key, pat in (("lds", r"\.group_segment_fixed_size:\s+(\d+)"), ("scratch", r"\.private_segment_fixed_size:\s+(\d+)"),
                         ("sgpr_spill", r"\.sgpr_spill_count:\s+(\d+)"), ("vgpr", r"\.vgpr_count:\s+(\d+)"),
                         ("vgpr_spill", r"\.vgpr_spill_count:\s+(\d+)"), ("name", r"\.name:\s+(\S+)")):
            m = re.match(r"\s+(?:- )?" + pat, raw)
            if m:
                cur[key] = m.group(1) if key == "name" else int(m.group(1))
        if "vgpr_spill" in cur and "name" in cur and "vgpr" in cur and "lds" in cur:
            table.append(cur)
            cur = {}
    return problems, table


def demangle_short(name):
    m = re.search(r"gibbs_sweeps_kernel(_big)?I(.*?)EEvNS", name)
    if not m:
        return name[:60]
    args = re.findall(r"L[ib](\d+)E", m.group(2))
    return "gibbs%s<%s>" % (m.group(1) or "", ",".join(args))


def main():
    paths = [a for a in sys.argv[1:] if not a.startswith("--")] or sorted(glob.glob(os.path.join(OBJ, "*gfx950*.s")))
    if not paths:
        print("isa_lint: no assembly found under %s (build with `make -C hmc.jl_amd/csrc`)" % OBJ)
        return 2
    bad = 0
    for p in paths:
        problems, table = lint_file(p, allow_phi="--allow-phi" in sys.argv)
        if "--table" in sys.argv:
            for k in table:
                print("%-34s vgpr %3d agpr %3d vspill %3d sspill %3d scratch %4d lds %6d" % (
                    demangle_short(k["name"]), k.get("vgpr", -1), k.get("agpr", 0), k.get("vgpr_spill", 0),
                    k.get("sgpr_spill", 0), k.get("scratch", 0), k.get("lds", 0)))
        if "--table" not in sys.argv:
            scr = [demangle_short(k["name"]) + ":%dB" % k["scratch"] for k in table if k.get("scratch", 0) > 0]
            if scr:
                print("warning: %s: scratch spills in %s" % (os.path.basename(p).split("-hip-")[0], " ".join(scr)))
        for path, n, kernel, what, text in problems:
            bad += 1
            print("%s:%d: [%s] %s: %s" % (os.path.basename(path), n, demangle_short(kernel or "?"), what, text))
    print("isa_lint: %d file(s), %d problem(s)" % (len(paths), bad))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
