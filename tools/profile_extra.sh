#!/bin/bash
# rocprofv3 kernel statistics of bench.py INCLUDING the extra shapes (configs[3] K=8, the 460-window production shape,
# 2048 windows, the chunked host entry): one stats CSV naming every kernel that ran.  Usage: tools/profile_extra.sh <tag>
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r02}
O=$R/gpurun_out/prof_extra_$TAG
mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$O/trace" -- python3 "$R/bench.py" --steps 5 --warmup 1 --no-cpu-baseline > "$O/trace.log" 2>&1; echo trace=$?
for f in $(find "$O/trace" -name "*kernel_stats.csv"); do cut -c1-260 "$f"; done
