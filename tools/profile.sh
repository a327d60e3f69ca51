#!/bin/bash
# rocprofv3 evidence for bench.py on the GPU box: kernel-trace stats, then PMC passes (each alone,
# never combined with sys/hip/hsa tracing).  Usage: tools/profile.sh <tag>   (run via gpurun)
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r01}
O=$R/gpurun_out/prof_$TAG
mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$O/trace" -- python3 "$R/bench.py" --no-cpu-baseline --no-extra > "$O/trace.log" 2>&1; echo trace=$?
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$O/pmc_fetch" -- python3 "$R/bench.py" --steps 3 --warmup 1 --no-cpu-baseline --no-extra > "$O/pmc_fetch.log" 2>&1; echo fetch=$?
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$O/pmc_write" -- python3 "$R/bench.py" --steps 3 --warmup 1 --no-cpu-baseline --no-extra > "$O/pmc_write.log" 2>&1; echo write=$?
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT --output-format csv -d "$O/pmc_sq" -- python3 "$R/bench.py" --steps 3 --warmup 1 --no-cpu-baseline --no-extra > "$O/pmc_sq.log" 2>&1; echo sq=$?
find "$O" -name "*.csv" | sed "s|$R/||"
for f in $(find "$O/trace" -name "*kernel_stats.csv"); do head -3 "$f" | cut -c1-220; done
for d in pmc_fetch pmc_write pmc_sq; do for f in $(find "$O/$d" -name "*counter_collection.csv"); do echo "== $d"; head -1 "$f"; grep gibbs "$f" | head -12 | cut -c1-400; done; done
