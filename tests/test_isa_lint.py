"""The shipped kernels' own assembly must be free of the allocator miscompile behind round 1's GPU memory fault
(DESIGN.md section 5a): `tools/isa_lint.py` on csrc/obj/*.s (kept by -save-temps when the library is built)."""
import glob
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OBJ = os.path.join(ROOT, "hmc.jl_amd", "csrc", "obj")

BAD = """\
_Z4kernv: ; @k
; %bb.0:
\tv_mov_b32_e32 v1, 0
.LBB0_229:
\tv_writelane_b32 v253, s36, 16
\tv_accvgpr_write_b32 a0, v10
\tv_accvgpr_write_b32 a1, v11
\tv_writelane_b32 v253, s45, 21
\ts_nop 1
\ts_or_b64 exec, exec, s[2:3]
\tv_mov_b32_e32 v7, 0
"""
GOOD = BAD.replace("\tv_accvgpr_write_b32 a0, v10\n\tv_accvgpr_write_b32 a1, v11\n", "").replace(
    "\tv_mov_b32_e32 v7, 0\n", "\tv_accvgpr_write_b32 a0, v10\n\tv_accvgpr_write_b32 a1, v11\n")
NARROW = """\
.LBB0_661:
\ts_mov_b64 s[2:3], exec
\tscratch_load_dwordx2 v[16:17], off, off offset:444
\ts_and_b64 s[0:1], s[2:3], s[0:1]
\ts_mov_b64 exec, s[0:1]
"""


# a PHI copy: the value is computed INSIDE the divergent region and copied for its lanes before the join -- harmless
PHI = """\
_Z4kernv: ; @k
; %bb.0:
\ts_and_saveexec_b64 s[2:3], vcc
\ts_cbranch_execz .LBB0_3
; %bb.2:
\tv_add_f64 v[10:11], v[4:5], v[6:7]
.LBB0_3:
\tv_mov_b64_e32 v[20:21], v[10:11]
\ts_or_b64 exec, exec, s[2:3]
\tv_mov_b32_e32 v7, 0
"""
# the same copy of a value that was written BEFORE the region: every lane holds one, only the region's lanes are copied
LIVEIN = PHI.replace("\tv_add_f64 v[10:11], v[4:5], v[6:7]\n", "\tv_add_f64 v[12:13], v[4:5], v[6:7]\n")


def run(path, *extra):
    return subprocess.run([sys.executable, os.path.join(ROOT, "tools", "isa_lint.py"), path] + list(extra), capture_output=True, text=True)


def test_lint_flags_the_round1_pattern(tmp_path):
    """The reduced form of the faulting build's block (one_256.s:1678-1697 in DESIGN.md 5a) is rejected, the
    corrected order and an exec-NARROWING sequence are accepted."""
    for name, text, rc in (("bad.s", BAD, 1), ("good.s", GOOD, 0), ("narrow.s", NARROW, 0)):
        p = tmp_path / name
        p.write_text(text)
        r = run(str(p))
        assert r.returncode == rc, (name, r.stdout)


def test_lint_classifies_hits_by_where_the_copied_value_was_written(tmp_path):
    """The region walk: a copy ahead of the exec restore whose source was written inside the region is 'phi-like' (refused by
    default -- the gate stays conservative -- and let through by --allow-phi); one whose source is live-in is the round-1 fault."""
    for name, text, kind in (("phi.s", PHI, "phi-like"), ("livein.s", LIVEIN, "live-in"), ("bad.s", BAD, "live-in")):
        p = tmp_path / name
        p.write_text(text)
        r = run(str(p))
        assert r.returncode == 1 and "[%s]" % kind in r.stdout, (name, r.stdout)
        r2 = run(str(p), "--allow-phi")
        assert r2.returncode == (0 if kind == "phi-like" else 1), (name, r2.stdout)


def test_shipped_build_is_lint_clean():
    files = sorted(glob.glob(os.path.join(OBJ, "*-hip-amdgcn-amd-amdhsa-gfx950.s")))
    if not files:
        pytest.skip("no compiler assembly under csrc/obj (library built elsewhere)")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "isa_lint.py")] + files, capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-4000:]
    assert len(files) >= 10
