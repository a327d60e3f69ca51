"""bench.py's multi-rank path on the one GPU this box has: `--gpus 2 --rehearse-shared-gpu` starts two
fresh ranks through torch.distributed.run (rendezvous on 127.0.0.1), both on cuda:0, process group gloo (RCCL refuses two
ranks on one device) -- every line of the N > 1 path except the backend name: global window ids per rank, the gather of
the summary blocks to rank 0, barrier + synchronize fences, MAX over ranks, rank 0's JSON line relayed by the parent.
The numbers of a rehearsal are never reported (`"rehearsal": true`)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.timeout(600)
def test_bench_two_ranks_on_the_shared_gpu(hmclib):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--rehearse-shared-gpu", "--steps", "2",
                        "--warmup", "1"], capture_output=True, text=True, env=env, timeout=540)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-2000:])
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{") and '"metric"' in ln]
    assert len(lines) == 1, r.stdout[-2000:]
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["rehearsal"] is True and rec["steps"] == 2 and rec["scaling"] == "weak"
    assert rec["config"]["windows_flagged"] == 0 and rec["config"]["windows_per_gpu"] == 256
    # whole-job aggregate over both ranks: 2 x 256 windows x 1000 draws x 2 steps in the MAX-over-ranks time
    assert abs(rec["value"] - 2 * 256 * 1000 / (rec["ms_per_step"] * 1e-3)) / rec["value"] < 1e-9
    assert rec["value"] > 1e6 and "extra" not in rec and "cpu_baseline" not in rec


@pytest.mark.timeout(300)
def test_bench_line_contract_on_one_gpu(hmclib):
    """`python bench.py --steps 3 --warmup 1` prints ONE JSON line with the contract's keys, the HBM-model roofline (live HIP-event
    kernel time), the VALU-issue roofline and internally consistent numbers (the driver checks the same things from outside)."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--no-extra"],
                       capture_output=True, text=True, timeout=240, env={k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")})
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
              "data", "config", "roofline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1 and d["dtype"] == "f64" and d["vs_baseline"] is None
    assert d["scaling"] == "weak" and d["higher_is_better"] is True and d["data"] == "synthetic" and "workload" in d["config"]
    assert abs(d["value"] - 256 * 1000 / (d["ms_per_step"] * 1e-3)) / d["value"] < 1e-9
    rf = d["roofline"]
    assert rf["bound"] == "hbm" and rf["unit"] == "GB/s" and rf["peak"] == 8000.0
    assert abs(rf["achieved"] - 58160 * 256 * 1000 / (rf["kernel_ms"] * 1e-3) / 1e9) / rf["achieved"] < 1e-9
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-12 and rf["kernel_ms"] <= d["ms_per_step"] * 1.001
    assert rf["kernel"].startswith("hmcg::gibbs_sweeps_kernel<3,4,256,false,false,4,2>") and rf["traffic"] is not None
    assert rf["measured_hbm_GBps"] < 0.01 * rf["model_GBps"]                  # the chain never leaves the chip
    ra = d["roofline_alu"]
    assert ra["bound"] == "valu_issue" and 0.3 < ra["frac"] < 1.0 and 2000 < ra["in_kernel_clock_mhz"] <= 2400
    assert 20e6 < d["value"] < 200e6
