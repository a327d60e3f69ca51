#!/usr/bin/env python
"""Headline benchmark: Gibbs draws/sec on synthetic 3-state Gaussian-HMM windows
(BASELINE.json configs[1]: K=3, T=1000, 256 windows, 1000 draws per GPU).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one batched estimate call (hmcg_estimate_batch_device) over this rank's 256
windows with inputs already resident in HBM: 0 burn-in + 1000 kept sweeps per window,
h=12 forecast per draw, on-device summary means; for N > 1 the step ends with the only
exchange of the path, the gather of the per-window summary blocks to rank 0 (RCCL).
Weak scaling: every GPU gets its own 256 windows (global window ids, so the sharded run
equals the unsharded one).  One JSON line is printed by rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

K, T, W_PER_GPU, DRAWS, HORIZON = 3, 1000, 256, 1000, 12
HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)


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
    window sized from a calibration run to ~8 s of wall time."""
    from oracle import oracle
    cores = usable_cores(oracle.max_threads())
    nwin = min(2 * cores, Y.shape[0])
    t0 = time.perf_counter()
    oracle.estimate_batch(Y[:1], Tw[:1], K, 0, 200, (HORIZON,), yreal[:1], nthreads=1)
    rate1 = 200 / (time.perf_counter() - t0)
    n = int(min(20000, max(200, 8.0 * rate1 * cores / nwin)))
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--threads-per-window", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--rehearse-shared-gpu", action="store_true",
                    help="multi-rank plumbing rehearsal on a one-GPU box: every rank uses cuda:0 and the process group "
                         "is gloo (RCCL refuses two ranks on one device); never used for reported numbers")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    import hmc_jl_amd
    from hmc_jl_amd import device as hdev, shard, synth

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit("--gpus %d needs torch.distributed.run with --nproc-per-node %d" % (args.gpus, args.gpus))
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
    fence()
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
        tm = panel.last_timing
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
                         "kernel": "hmcg::gibbs_sweeps_kernel<3,%d,%d,false,false,%d,2>" % (tm.steps_per_thread, tm.threads_per_window, tm.helper_waves),
                         "kernel_ms": k_ms, "algorithmic_bytes_per_launch": B * W_PER_GPU * DRAWS,
                         "note": "algorithmic bytes = 58160 B/draw (SURVEY 8d) x 256 windows x 1000 draws; the chain "
                                 "state is register/LDS-resident, so the physical limiter is fp64 VALU latency, not HBM"},
        }
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(Y, Tw, yreal)
        print(json.dumps(line))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
