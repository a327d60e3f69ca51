"""Condense a tools/profile_k8.sh output directory (gpurun_out/prof_k8_<tag>) into profiles/<round>/pmc_k8.json plus the
kernel-stats CSV and the head of each counter pass.  Usage: python tools/summarize_profile_k8.py gpurun_out/prof_k8_r03 profiles/r03 [suffix]"""
import csv
import glob
import json
import os
import shutil
import sys

src, dst = sys.argv[1], sys.argv[2]
sfx = sys.argv[3] if len(sys.argv) > 3 else ""
os.makedirs(dst, exist_ok=True)
K, T, W, SWEEPS = 8, 5000, 512, 1000
ALGO = (T * (8 + 16 * K + 2) + 8 * (3 * K + K * K + 2)) * W * SWEEPS        # 690 720 B/draw
L = (T + 255) // 256
PDF_SCRATCH = 3 * 8 * L * K * 256 * W * SWEEPS                                # fscr[W][L][K/2][NT][2]: written once, read twice per sweep (product, replay) since round 4


def rows(pattern):
    out = []
    for f in sorted(glob.glob(os.path.join(src, pattern), recursive=True)):
        with open(f, newline="") as fh:
            out += list(csv.DictReader(fh))
    return out


name = "gibbs_sweeps_kernel"
stats = [r for r in rows("trace/**/*kernel_stats.csv") if name in r["Name"]]
for f in glob.glob(os.path.join(src, "trace/**/*kernel_stats.csv"), recursive=True):
    shutil.copy(f, os.path.join(dst, "k8%s_kernel_stats.csv" % sfx))
out = {"command": "tools/profile_k8.sh: rocprofv3 --kernel-trace [--stats | --pmc <one counter group>] --output-format csv -- "
                  "python3 tools/bench_cfg.py 8 5000 512 1000 1 (two launches per pass); condensed by tools/summarize_profile_k8.py",
       "kernel": stats[0]["Name"] if stats else None,
       "rocprof_kernel_avg_ms": float(stats[0]["AverageNs"]) / 1e6 if stats else None,
       "rocprof_kernel_calls": int(stats[0]["Calls"]) if stats else None}
cnt = {}
for tag in ("pmc_fetch", "pmc_write", "pmc_sq", "pmc_sq2", "pmc_grbm", "pmc_tcc", "pmc_ea"):
    rr = [r for r in rows(tag + "/**/*counter_collection.csv") if name in r["Kernel_Name"]]
    for f in glob.glob(os.path.join(src, tag + "/**/*counter_collection.csv"), recursive=True):
        with open(f) as fh, open(os.path.join(dst, "k8%s_%s_head.csv" % (sfx, tag)), "w") as o:
            for i, line in enumerate(fh):
                if i < 20:
                    o.write(line)
    for r in rr:
        cnt.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
        out.setdefault("registers", {"VGPR_Count": r["VGPR_Count"], "Accum_VGPR_Count": r["Accum_VGPR_Count"], "SGPR_Count": r["SGPR_Count"],
                                     "LDS_Block_Size": r["LDS_Block_Size"], "Scratch_Size": r["Scratch_Size"], "Workgroup_Size": r["Workgroup_Size"]})
c = {k: sum(v) / len(v) for k, v in cnt.items()}
out["counters_per_launch"] = c
if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
    fetch = c["FETCH_SIZE"] * 1024.0 * 2.0        # KB; gfx950 reports half the bytes of wide coalesced reads (MI355X_MICROARCH.md, HBM)
    write = c["WRITE_SIZE"] * 1024.0
    out.update({"fetch_bytes_corrected": fetch, "write_bytes": write, "hbm_bytes_per_launch": fetch + write,
                "algorithmic_bytes_per_launch": ALGO, "pdf_scratch_bytes_per_launch_expected": PDF_SCRATCH,
                "correction": "FETCH_SIZE (KB) x 1024 x 2: on gfx950 FETCH_SIZE counts 128-B requests at 64 B (guide, HBM section); the pdf "
                              "scratch is read with 16 B/lane coalesced loads (1 KB per wave-instruction), twice per sweep since round 4 "
                              "(chunk product and replay; written once): the doubled figure should be about twice the written bytes. WRITE_SIZE (KB) x 1024 is exact for streaming stores. Fabric-side counters: "
                              "Infinity-Cache hits are included."})
    if out["rocprof_kernel_avg_ms"]:
        out["fabric_GBps"] = (fetch + write) / (out["rocprof_kernel_avg_ms"] * 1e-3) / 1e9
        out["algorithmic_GBps"] = ALGO / (out["rocprof_kernel_avg_ms"] * 1e-3) / 1e9
if "SQ_WAVE_CYCLES" in c:
    wc = c["SQ_WAVE_CYCLES"]
    waves = c.get("SQ_WAVES", 2048.0)
    d = {"note": "SQ_* cycle counters tick in quad-cycles (x4 = shader cycles)",
         "wave_cycles_per_window_sweep": wc * 4.0 / waves / SWEEPS,
         "valu_insts_per_wave_per_sweep": c.get("SQ_INSTS_VALU", 0.0) / waves / SWEEPS,
         "salu_insts_per_wave_per_sweep": c.get("SQ_INSTS_SALU", 0.0) / waves / SWEEPS,
         "lds_insts_per_wave_per_sweep": c.get("SQ_INSTS_LDS", 0.0) / waves / SWEEPS,
         "wait_any_frac": c.get("SQ_WAIT_ANY", 0.0) / wc, "active_any_frac": c.get("SQ_ACTIVE_INST_ANY", 0.0) / wc,
         "valu_issue_frac_at_4_cycles": c.get("SQ_INSTS_VALU", 0.0) * 4.0 / (wc * 4.0)}
    for k in ("SQ_WAIT_INST_ANY", "SQ_WAIT_INST_LDS", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "SQ_LDS_BANK_CONFLICT"):
        if k in c:
            d[k.lower() + "_frac"] = c[k] / wc
    if "SQ_INSTS_VMEM_RD" in c:
        d["vmem_rd_insts_per_wave_per_sweep"] = c["SQ_INSTS_VMEM_RD"] / waves / SWEEPS
        d["vmem_wr_insts_per_wave_per_sweep"] = c["SQ_INSTS_VMEM_WR"] / waves / SWEEPS
    out["derived"] = d
if "GRBM_GUI_ACTIVE" in c and out["rocprof_kernel_avg_ms"]:
    out["effective_clock_mhz_grbm"] = c["GRBM_GUI_ACTIVE"] / 8.0 / (out["rocprof_kernel_avg_ms"] * 1e-3) / 1e6
json.dump(out, open(os.path.join(dst, "pmc_k8%s.json" % sfx), "w"), indent=1)
print(json.dumps(out, indent=1))
