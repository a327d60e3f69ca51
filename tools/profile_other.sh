set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/prof_other
rm -rf "$O"; mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$O/w2048" -- python3 "$R/tools/bench_cfg.py" 3 1000 2048 1000 > "$O/w2048.log" 2>&1; echo w2048=$?
rocprofv3 --kernel-trace --stats --output-format csv -d "$O/sig" -- python3 "$R/tools/bench_signals.py" > "$O/sig.log" 2>&1; echo sig=$?
rocprofv3 --kernel-trace --stats --output-format csv -d "$O/k8" -- python3 "$R/tools/bench_cfg.py" 8 5000 512 1000 1 > "$O/k8.log" 2>&1; echo k8=$?
for d in w2048 sig k8; do for f in $(find "$O/$d" -name "*kernel_stats.csv"); do echo "== $d"; head -2 "$f" | cut -c1-260; done; done
