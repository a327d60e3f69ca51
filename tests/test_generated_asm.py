"""The two generated assembly loops of the K = 8 kernel (csrc/product_asm_k8.inc, csrc/replay_asm_k8.inc) are what their
generators produce today: an edit of a generator without regenerating (or a hand edit of an .inc) fails here, on the CPU."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("gen,inc", [("gen_product_asm.py", "product_asm_k8.inc"), ("gen_replay_asm.py", "replay_asm_k8.inc")])
def test_generated_include_is_current(tmp_path, gen, inc):
    out = tmp_path / inc
    env = {k: v for k, v in os.environ.items() if k not in ("PRODUCT_NSR", "REPLAY_EPS_FAST")}
    subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", gen), "--out", str(out)], env=env, stdout=subprocess.DEVNULL)
    assert out.read_text() == open(os.path.join(ROOT, "hmc.jl_amd", "csrc", inc)).read()


def test_asm_statements_pad_their_hazards():
    """What the compiler's hazard recogniser would do and an asm statement must do by itself: wait
    states before the first VMEM use of an SGPR operand, and no VALU read of a v_rcp_f64 result by the next instruction."""
    for inc in ("product_asm_k8.inc", "replay_asm_k8.inc"):
        lines = [ln.strip().strip('"').replace("\\n", "") for ln in open(os.path.join(ROOT, "hmc.jl_amd", "csrc", inc)) if ln.strip().startswith('"')]
        assert lines[0] == "s_nop 4" and lines[-1] == "s_nop 4", inc
        for i, ln in enumerate(lines[:-1]):
            if ln.startswith("v_rcp_f64"):
                dst = ln.split()[1].rstrip(",")
                assert dst not in lines[i + 1], (inc, ln, lines[i + 1])
