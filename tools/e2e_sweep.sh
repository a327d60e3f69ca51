#!/bin/bash
export HMCG_DIAG=1   # arms the library's diagnostic switches (read once at first use)
# host-entry call time (plain-C caller, median / min of 24 calls) under chunk-schedule settings: gpurun -- 'bash tools/e2e_sweep.sh'
export TRACE_CALLS=24
for cfg in "" "HMCG_NO_DIRECT_TAIL=1" "HMCG_CHUNK_KEEP=1/2" "HMCG_CHUNK_KEEP=3/5" "HMCG_CHUNK_FLOOR_DIV=64" "HMCG_CHUNK_FLOOR_DIV=16" "HMCG_CHUNK_KEEP=2/3 HMCG_CHUNK_FLOOR_DIV=64" "HMCG_CHUNK_KEEP=3/4 HMCG_CHUNK_FLOOR_DIV=64"; do
  for rep in 1 2; do
    echo -n "[$cfg] "; env $cfg python tools/trace_host_entry.py 2>/dev/null | grep "cdriver bench" | sed 's/(W=.*timed call//'
  done
done
