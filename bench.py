#!/usr/bin/env python
"""Headline benchmark: Gibbs draws/sec on synthetic 3-state Gaussian-HMM windows
(BASELINE.json configs[1]: K=3, T=1000, 256 windows, 1000 draws per GPU).

    python bench.py --gpus N --steps K --warmup W

A "step" is one batched estimate call (hmcg_estimate_batch_device) over this rank's 256
windows with inputs already resident in HBM: 0 burn-in + 1000 kept sweeps per window,
h=12 forecast per draw, on-device summary means; for N > 1 the step ends with the only
exchange of the path, the gather of the per-window summary blocks to rank 0 (RCCL).
Weak scaling: every GPU gets its own 256 windows (global window ids, so the sharded run
equals the unsharded one).  One JSON line is printed by rank 0.

Launch: with N > 1 and no WORLD_SIZE in the environment this process touches no GPU; it
starts `python -m torch.distributed.run --nproc-per-node N` on this same file as a child
(one rank per GPU) and relays rank 0's JSON line.  Under torchrun (WORLD_SIZE set) it is a rank.

At N = 1 the line also carries `extra`: the other shapes of BASELINE.json and the end-to-end
(host buffers in, all per-draw outputs out) rate of the headline shape, each measured after the
headline timed region.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

K, T, W_PER_GPU, DRAWS, HORIZON = 3, 1000, 256, 1000, 12
HBM_PEAK_GBS = 8000.0         # MI355X_MICROARCH.md: 8.0 TB/s spec
HBM_ACHIEVABLE_GBS = 6300.0   # ... ~6.3 TB/s measured copy bandwidth (SURVEY.md 8d asks for both fractions)


def algorithmic_bytes_per_draw(T_, K_, H_):
    """read Y, write+read pif, write+read X, per-draw outputs."""
    return T_ * (8 + 16 * K_ + 2) + 8 * (3 * K_ + K_ * K_ + 2 * H_)


def usable_cores(omp_max):
    """Threads this process may actually run at once: min(OpenMP's view, the affinity mask,
    the cgroup CPU quota) -- the GPU box exposes every host core but caps the job's share."""
    n = omp_max
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except AttributeError:
        pass
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(float(quota) / float(period) + 0.5)))
    except (OSError, ValueError):
        pass
    return max(1, n)


def cpu_baseline(Y, Tw, yreal):
    """The oracle (kind "port": C restatement of the reference's CPU path) timed on the
    host cores over a bounded sample of the same workload: 2 windows per core, draws per
    window sized from a single-thread calibration run to ~12 s of wall time (plus ~1.5 s for the
    "faithful cost" variant)."""
    from oracle import oracle
    cores = usable_cores(oracle.max_threads())
    nwin = min(2 * cores, Y.shape[0])
    t0 = time.perf_counter()
    oracle.estimate_batch(Y[:1], Tw[:1], K, 0, 200, (HORIZON,), yreal[:1], nthreads=1)
    rate1 = 200 / (time.perf_counter() - t0)
    n = int(min(400000, max(200, 12.0 * rate1 * cores / nwin)))
    t0 = time.perf_counter()
    oracle.estimate_batch(Y[:nwin], Tw[:nwin], K, 0, n, (HORIZON,), yreal[:nwin], nthreads=cores)
    dt = time.perf_counter() - t0
    out = {"value": nwin * n / dt, "unit": "Gibbs draws/s", "cores": cores, "kind": "port",
           "sample": "%d windows x %d draws (K=3,T=1000) on %d OpenMP threads, %.1f s; oracle/hmc_oracle.c "
                     "without the reference's accidental O(T^2) index search" % (nwin, n, cores, dt)}
    nf = max(50, n // 8)
    t0 = time.perf_counter()
    oracle.estimate_batch(Y[:nwin], Tw[:nwin], K, 0, nf, (HORIZON,), yreal[:nwin], nthreads=cores, faithful_cost=True)
    dtf = time.perf_counter() - t0
    out["faithful_cost_value"] = nwin * nf / dtf
    out["faithful_cost_note"] = ("same arithmetic, plus the reference's per-step linear search (src/Hmc.jl:409) and "
                                 "K^2 pdf evaluations (:415): %d windows x %d draws, %.1f s" % (nwin, nf, dtf))
    out["single_thread_value"] = rate1
    return out


def kernel_name(K_, tm, sig=False, smooth=False):
    """The instantiation that ran, as rocprofv3 names it (template arguments from the call's own timing record)."""
    if tm.occupancy == 0:          # the LDS-resident kernel: <K, NT, SM, STREAM, SIG>
        return "hmcg::gibbs_sweeps_kernel_big<%d,%d,%s,%s,%s>" % (K_, tm.threads_per_window, str(smooth).lower(),
                                                                  str(bool(tm.streaming)).lower(), str(sig).lower())
    return "hmcg::gibbs_sweeps_kernel<%d,%d,%d,%s,%s,%d,%d>" % (
        K_, tm.steps_per_thread, tm.threads_per_window, str(sig).lower(), str(smooth).lower(), tm.helper_waves, tm.occupancy)


def shape_record(name, K_, lens, draws, reps=3, device=0):
    """One device-resident call of another BASELINE shape: kernel time from HIP events on the launch stream."""
    import numpy as np
    from hmc_jl_amd import device as hdev, synth
    W = len(lens)
    Tmax = int(max(lens))
    ragged = None if len(set(lens)) == 1 else list(lens)
    Y, Tw, fut = synth.generate_panel(W, Tmax, K_, horizon_pad=HORIZON, ragged=ragged)
    panel = hdev.DevicePanel(Y, Tw, K_, draws, (HORIZON,), fut[:, HORIZON - 1:HORIZON], device=device, keep_draws=True)
    panel.run(burnin=0)
    ms = [panel.run(burnin=0) for _ in range(reps)]
    tm = panel.last_timing
    k_ms = float(np.mean(ms))
    by = float(sum(algorithmic_bytes_per_draw(int(t), K_, 1) for t in lens)) * draws
    ach = by / (k_ms * 1e-3) / 1e9
    rec = {"workload": name, "K": K_, "windows": W, "T_min": int(min(lens)), "T_max": Tmax, "draws_per_window": draws,
           "kernel": kernel_name(K_, tm), "kernel_ms": k_ms, "value": W * draws / (k_ms * 1e-3), "unit": "Gibbs draws/s",
           "algorithmic_bytes_per_launch": by, "achieved_GBps": ach, "frac_of_8000": ach / HBM_PEAK_GBS,
           "frac_of_6300": ach / HBM_ACHIEVABLE_GBS, "steps_per_thread": tm.steps_per_thread,
           "helper_waves": tm.helper_waves, "lds_bytes_per_window": tm.lds_bytes, "length_buckets": tm.buckets,
           "windows_flagged": int((panel.status != 0).sum().item())}
    if tm.buckets > 1:
        # the same call as ONE launch sized for the longest window (no min_T hint: what every call did until round 3)
        panel.run(burnin=0, bucketed=False)
        one = float(np.mean([panel.run(burnin=0, bucketed=False) for _ in range(reps)]))
        rec["single_launch_kernel_ms"] = one
        rec["single_launch_value"] = W * draws / (one * 1e-3)
        rec["kernel"] += " (+ %d shorter length classes side by side)" % (tm.buckets - 1)
    del panel
    return rec


def c_caller_record(Y, Tw, yreal, reps=20):
    """The same workload from a plain-C caller (tests/cdriver, no Python in the process, buffers allocated once and reused --
    what a Julia `ccall` site does): median wall time per hmcg_estimate_batch call (mean, min and max beside it)."""
    import struct
    import tempfile
    import numpy as np
    drv = os.path.join(ROOT, "tests", "cdriver", "hmcg_cdriver")
    if not os.path.exists(drv):
        return {"error": "tests/cdriver/hmcg_cdriver not built"}
    W, ld = Y.shape
    hd = [0x484d4347, 3, W, K, ld, 0, DRAWS, 1, HORIZON, 0, 0, 0, 0, 0, 0, 0, 0, 0, reps, 0]
    with tempfile.TemporaryDirectory() as tmp:
        req = os.path.join(tmp, "req.bin")
        with open(req, "wb") as f:
            f.write(struct.pack("<20i", *hd))
            f.write(struct.pack("<3d", 0.0, 0.0, 0.0))
            f.write(np.ascontiguousarray(Y, dtype="<f8").tobytes())
            f.write(np.ascontiguousarray(Tw, dtype="<i4").tobytes())
            f.write(np.ascontiguousarray(yreal, dtype="<f8").tobytes())
        r = subprocess.run([drv, req, os.path.join(tmp, "resp.bin")], capture_output=True, text=True, timeout=300,
                           env={k: v for k, v in os.environ.items() if k != "LD_PRELOAD"})
    import re
    m = re.search(r"cdriver bench: ([0-9.]+) ms per call.*?(\d+) launches, kernels ([0-9.]+) ms, call ([0-9.]+) ms; median of calls, mean ([0-9.]+) min ([0-9.]+) max ([0-9.]+)", r.stdout)
    if r.returncode != 0 or not m:
        return {"error": (r.stdout + r.stderr)[-300:]}
    ms = float(m.group(1))
    return {"ms_per_call": ms, "value": W * DRAWS / (ms * 1e-3), "unit": "Gibbs draws/s", "launches": int(m.group(2)),
            "kernel_ms_sum_timed_call": float(m.group(3)), "calls": reps, "statistic": "median of the per-call wall times",
            "mean_ms": float(m.group(5)), "min_ms": float(m.group(6)), "max_ms": float(m.group(7))}


def end_to_end_record(Y, Tw, yreal, reps=7):
    """SURVEY.md 8(d)'s wall-clock definition on the headline shape: pageable host arrays in (H2D of the Y panel), every
    per-draw output back in the caller's pageable arrays (41 MB), through hmcg_estimate_batch."""
    import numpy as np
    from hmc_jl_amd import _lib
    out = None
    for _ in range(3):
        out = _lib.estimate_batch_host(Y, Tw, K, 0, DRAWS, (HORIZON,), yreal, out=out)
    t, lib_ms = [], []
    for _ in range(reps):
        t0 = time.perf_counter()
        out = _lib.estimate_batch_host(Y, Tw, K, 0, DRAWS, (HORIZON,), yreal, out=out)
        t.append(time.perf_counter() - t0)
        lib_ms.append(out["call_ms"])
    ms = float(np.median(t)) * 1e3
    r = out
    t2 = []
    for _ in range(reps):
        t0 = time.perf_counter()
        _lib.estimate_batch_host(Y, Tw, K, 0, DRAWS, (HORIZON,), yreal, want_draws=False)
        t2.append(time.perf_counter() - t0)
    ms2 = float(np.median(t2)) * 1e3
    nbytes = sum(r[k].nbytes for k in ("mu", "sig2", "A", "pi_end", "fcast", "summary", "status"))
    rec = {"workload": "configs[1] through the host entry: H2D of Y, %d chunked launches, D2H of all per-draw outputs "
                       "(%.1f MB) into pageable caller arrays (reused from call to call)" % (r["launches"], nbytes / 1e6),
           "ms_per_call": ms, "value": W_PER_GPU * DRAWS / (ms * 1e-3), "unit": "Gibbs draws/s",
           "library_call_ms": float(np.median(lib_ms)), "kernel_ms_sum": r["kernel_ms"], "launches": r["launches"],
           "summary_only_ms_per_call": ms2, "summary_only_value": W_PER_GPU * DRAWS / (ms2 * 1e-3),
           "note": "ms_per_call is the Python caller's wall time (ctypes wrapper included); library_call_ms the wall time inside "
                   "hmcg_estimate_batch; c_caller is the same call from tests/cdriver (plain C, no Python)"}
    rec["c_caller"] = c_caller_record(Y, Tw, yreal)
    return rec


def sustained_record(panel, ids0, seconds=2.5):
    """Back-to-back launches of the headline shape for >= `seconds` (the 20 timed steps of the headline region last 0.1 s:
    they say nothing about the clock the chip holds under this kernel for seconds; MI355X_MICROARCH.md, DVFS give-back).
    Launches are enqueued untimed on the library stream; wall time between two device synchronisations."""
    import torch
    one = panel.run(burnin=0, seed=1234, window_base=ids0, timed=True)
    n = max(20, int(seconds * 1e3 / one) + 1)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        panel.run(burnin=0, seed=1234, window_base=ids0, timed=False)
    panel.sync()
    dt = time.perf_counter() - t0
    last = panel.run(burnin=0, seed=1234, window_base=ids0, timed=True)      # HIP-event time of the launch right after
    return {"launches": n, "seconds": dt, "ms_per_launch": dt / n * 1e3, "value": W_PER_GPU * DRAWS * n / dt,
            "unit": "Gibbs draws/s", "kernel_ms_after": last, "first_kernel_ms": one,
            "note": "%d launches enqueued back to back, wall clock between synchronisations (launch gaps included)" % n}


def alu_roofline(kernel_ms, key):
    """The roofline that actually binds these kernels: VALU issue.  A SIMD issues one wave-instruction of a lone wave per
    4 shader cycles (fp64 FMA included: tools/ubench/clock.hip measures 4.03 cycles per fp64 wave-op with every CU busy,
    i.e. the 78.6 TFLOP/s vector peak at 2.4 GHz), so a launch cannot take less than
        VALU wave-instructions / (CUs x 4 SIMDs) x 4 cycles / clock.
    Instruction counts come from the SQ_INSTS_VALU pass kept under profiles/ (same launch shape), the clock from the
    in-kernel s_memtime / s_memrealtime ratio of the stamped build after >= 2 s of launches (profiles/clock.json)."""
    try:
        ck = json.load(open(os.path.join(ROOT, "profiles", "clock.json")))[key]
    except Exception:
        return None
    insts, mhz = ck.get("valu_wave_insts_per_launch"), ck.get("clock_mhz")
    if not insts or not mhz:
        return None
    simds = 256 * 4
    bound_ms = insts / simds * 4.0 / (mhz * 1e6) * 1e3
    fp64 = ck.get("fp64_flop_per_launch")
    out = {"bound": "valu_issue", "valu_wave_insts_per_launch": insts, "in_kernel_clock_mhz": mhz,
           "cycles_per_wave_inst": 4.0, "issue_bound_ms": bound_ms, "kernel_ms": kernel_ms, "frac": bound_ms / kernel_ms,
           "source": ck.get("source")}
    if fp64:
        tf = fp64 / (kernel_ms * 1e-3) / 1e12
        out.update({"fp64_TFLOPs": tf, "fp64_vector_peak_TFLOPs": 78.6, "frac_of_fp64_peak": tf / 78.6})
    return out


def spawn_ranks(args):
    """Parent of an N-GPU run: no torch.cuda, no HIP call here -- the ranks are fresh child processes."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    proc = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    line = None
    for ln in proc.stdout.splitlines():
        if ln.startswith("{") and '"metric"' in ln:
            line = ln
    if line:
        print(line, flush=True)
    else:
        sys.stderr.write(proc.stdout)
    return proc.returncode if proc.returncode else (0 if line else 1)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--threads-per-window", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the extra shapes (profiling runs of the headline kernel)")
    ap.add_argument("--rehearse-shared-gpu", action="store_true",
                    help="multi-rank plumbing rehearsal on a one-GPU box: every rank uses cuda:0 and the process group "
                         "is gloo (RCCL refuses two ranks on one device); never used for reported numbers")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args))

    import numpy as np
    import torch
    import torch.distributed as dist
    import hmc_jl_amd  # noqa: F401
    from hmc_jl_amd import device as hdev, shard, synth

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit("--gpus %d under a launcher with WORLD_SIZE=%d" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: hmc.jl_amd has no CPU fallback")
    if args.rehearse_shared_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.rehearse_shared_gpu:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    W_total = W_PER_GPU * world
    ids = list(range(rank * W_PER_GPU, (rank + 1) * W_PER_GPU))
    Y, Tw, fut = synth.generate_panel(W_PER_GPU, T, K, horizon_pad=HORIZON, window_base=ids[0])
    yreal = fut[:, HORIZON - 1:HORIZON]
    panel = hdev.DevicePanel(Y, Tw, K, DRAWS, (HORIZON,), yreal, device=local_rank)

    def step():
        ms = panel.run(burnin=0, seed=1234, window_base=ids[0], threads_per_window=args.threads_per_window, timed=True)
        if world > 1:
            shard.gather_blocks(panel.summary, ids, W_total)
        return ms

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    if panel.last_timing is None:       # (--warmup 0: the launch geometry reported below comes from a timed launch)
        step()
    tm_geom = panel.last_timing
    st = None
    if world == 1:
        st = torch.cuda.Stream(device=local_rank)      # (a stream of its own: the library takes stream 0 to mean "use your own")
        # first use of a stream is not free (~5 ms): one launch on it outside the timed region, whatever --warmup says
        panel.run(burnin=0, seed=1234, window_base=ids[0], threads_per_window=args.threads_per_window, timed=False, stream=st.cuda_stream)
    fence()
    if world == 1:
        # one GPU: the K steps are enqueued back to back on one stream (no host round trip between steps: nothing on the host
        # consumes a step's result before the next one starts) and bracketed by HIP events ON THAT STREAM -- the kernel's
        # average launch duration over the timed region, launch gaps included -- and by the wall clock between the fences
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        e0.record(st)
        for _ in range(args.steps):
            panel.run(burnin=0, seed=1234, window_base=ids[0], threads_per_window=args.threads_per_window, timed=False,
                      stream=st.cuda_stream)
        e1.record(st)
        fence()
        dt = time.perf_counter() - t0
        kms = [e0.elapsed_time(e1) / args.steps]
    else:
        t0 = time.perf_counter()
        kms = [step() for _ in range(args.steps)]
        fence()
        dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device="cpu" if args.rehearse_shared_gpu else "cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    bad = int((panel.status != 0).sum().item())

    if rank == 0:
        value = W_total * DRAWS * args.steps / dt
        B = algorithmic_bytes_per_draw(T, K, 1)
        k_ms = float(np.mean(kms))
        achieved = B * W_PER_GPU * DRAWS / (k_ms * 1e-3) / 1e9
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath):
            try:
                traffic = json.load(open(tpath)).get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        tm = tm_geom
        line = {
            "metric": "Gibbs draws/sec (whole node), 3-state T=1000 x256 windows per GPU",
            "value": value, "unit": "Gibbs draws/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "rehearsal": bool(args.rehearse_shared_gpu),
            "config": {"workload": "configs[1]: 3-state Gaussian HMM, T=1000, 256 windows per GPU, 0 burn-in + 1000 "
                                   "draws per window per step, h=12 forecast per draw, on-device summary means",
                       "K": K, "T": T, "windows_per_gpu": W_PER_GPU, "draws_per_window": DRAWS,
                       "parallelism": "windows sharded over %d GPU(s), no data-path collective; summary gather to rank 0" % world,
                       "threads_per_window": tm.threads_per_window, "steps_per_thread": tm.steps_per_thread,
                       "helper_waves": tm.helper_waves, "lds_bytes_per_window": tm.lds_bytes, "windows_flagged": bad},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "peak_achievable": HBM_ACHIEVABLE_GBS, "frac_of_achievable": achieved / HBM_ACHIEVABLE_GBS,
                         "kernel": kernel_name(K, tm),
                         "kernel_ms": k_ms, "algorithmic_bytes_per_launch": B * W_PER_GPU * DRAWS,
                         "model_GBps": achieved,
                         "measured_hbm_GBps": (traffic / (k_ms * 1e-3) / 1e9) if traffic else None,
                         "note": "algorithmic bytes = 58160 B/draw (SURVEY 8d) x 256 windows x 1000 draws; the chain "
                                 "state is register/LDS-resident, so the physical limiter is fp64 VALU issue, not HBM"},
        }
        ra = alu_roofline(k_ms, "headline")
        if ra:
            line["roofline_alu"] = ra
        if world == 1 and not args.no_extra:
            extra = {}
            try:
                extra["sustained"] = sustained_record(panel, ids[0])
            except Exception as e:
                extra["sustained_error"] = repr(e)
            del panel
            try:
                e2e = end_to_end_record(Y, Tw, yreal)
                extra["end_to_end_host_entry"] = e2e
                # SURVEY 8(d)'s wall-clock definition (H2D of Y + kernels + D2H of every per-draw output) beside `value`,
                # which the bench contract defines with the inputs resident in HBM
                cc = e2e.get("c_caller", {})
                ms_e2e = cc.get("ms_per_call", e2e["ms_per_call"])
                line["end_to_end"] = {"value": W_PER_GPU * DRAWS / (ms_e2e * 1e-3), "unit": "Gibbs draws/s", "ms_per_call": ms_e2e,
                                      "caller": "plain C (tests/cdriver)" if "ms_per_call" in cc else "python ctypes",
                                      "device_resident_ms": k_ms, "overhead_vs_device_resident": ms_e2e / k_ms - 1.0,
                                      "definition": "SURVEY 8(d): pageable host Y in, all per-draw outputs (41 MB) back in the caller's arrays"}
                extra["cfg4_k8_T5000_w512"] = shape_record("configs[3]: 8-state, T=5000, 512 windows, 1000 draws", 8, [5000] * 512, 1000, reps=2)
                ra8 = alu_roofline(extra["cfg4_k8_T5000_w512"]["kernel_ms"], "k8")
                if ra8:
                    extra["cfg4_k8_T5000_w512"]["roofline_alu"] = ra8
                extra["production_460_expanding"] = shape_record(
                    "the reference's production shape (code/run_hmm.jl:79-109): 460 expanding windows T=120..579, K=3, 1000 draws",
                    3, list(range(120, 580)), 1000)
                extra["w2048_T1000"] = shape_record("configs[2] on one GPU: 3-state, T=1000, 2048 windows, 1000 draws", 3, [1000] * 2048, 1000)
            except Exception as e:                      # the headline line must still be printed
                extra["error"] = repr(e)
            line["extra"] = extra
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(Y, Tw, yreal)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
