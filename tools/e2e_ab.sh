#!/bin/bash
export HMCG_DIAG=1   # arms the library's diagnostic switches (read once at first use)
# End-to-end (host entry) call time of the headline shape under different pipeline settings, same box, interleaved:
#   gpurun -- 'bash tools/e2e_ab.sh "HMCG_SCATTER_THREADS=0" "HMCG_SCATTER_THREADS=3" "HMCG_CHUNK_FLOOR_DIV=32"'
for rep in 1 2; do
for setting in "$@"; do
env $setting timeout -k 10 300 python bench.py --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); h=j['extra']['end_to_end_host_entry']; print('$setting', 'device', round(j['ms_per_step'],3), 'py', round(h['ms_per_call'],3), 'lib', round(h['library_call_ms'],3), 'launches', h['launches'], 'C', h['c_caller']['ms_per_call'], h['c_caller']['min_ms'], h['c_caller']['max_ms'])"
done
done
