#!/bin/bash
# rocprofv3 evidence for the LDS-resident large-K kernel at configs[3] (8 states, T=5000, 512 windows, 1000 draws): kernel
# trace, then the PMC passes, one counter group per run (never combined with sys/hip/hsa tracing).
# Usage: tools/profile_k8.sh <tag>   (run via gpurun; results under gpurun_out/prof_k8_<tag>/)
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r03}
O=$R/gpurun_out/prof_k8_$TAG
mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
CMD="python3 $R/tools/bench_cfg.py 8 5000 512 1000 1"
rocprofv3 --kernel-trace --stats --output-format csv -d "$O/trace" -- $CMD > "$O/trace.log" 2>&1; echo trace=$?
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$O/pmc_fetch" -- $CMD > "$O/pmc_fetch.log" 2>&1; echo fetch=$?
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$O/pmc_write" -- $CMD > "$O/pmc_write.log" 2>&1; echo write=$?
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT --output-format csv -d "$O/pmc_sq" -- $CMD > "$O/pmc_sq.log" 2>&1; echo sq=$?
rocprofv3 --kernel-trace --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_SMEM SQ_BUSY_CYCLES --output-format csv -d "$O/pmc_sq2" -- $CMD > "$O/pmc_sq2.log" 2>&1; echo sq2=$?
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE --output-format csv -d "$O/pmc_grbm" -- $CMD > "$O/pmc_grbm.log" 2>&1; echo grbm=$?
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d "$O/pmc_tcc" -- $CMD > "$O/pmc_tcc.log" 2>&1; echo tcc=$?
rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum --output-format csv -d "$O/pmc_ea" -- $CMD > "$O/pmc_ea.log" 2>&1; echo ea=$?
find "$O" -name "*.csv" | sed "s|$R/||"
for f in $(find "$O/trace" -name "*kernel_stats.csv"); do head -4 "$f" | cut -c1-220; done
for d in pmc_fetch pmc_write pmc_sq pmc_sq2 pmc_grbm pmc_tcc pmc_ea; do for f in $(find "$O/$d" -name "*counter_collection.csv"); do echo "== $d"; head -1 "$f"; grep gibbs "$f" | head -24 | cut -c1-400; done; done
