#!/bin/bash
# A/B of the LDS-resident kernel's two forward filters on the configs[3] shape within one library build:
# HMCG_BIG_FWD=0 (per-lane chunk products) against the default (row-split).  gpurun -- 'bash tools/ab_fwd.sh'
mkdir -p gpurun_out; : > gpurun_out/ab_fwd.log
for i in 1 2; do
  for v in 0 1; do
    HMCG_BIG_FWD=$v timeout -k 10 200 python tools/bench_cfg.py 8 5000 512 1000 2 >> gpurun_out/ab_fwd.log 2>&1 || exit 1
  done
done
cat gpurun_out/ab_fwd.log
