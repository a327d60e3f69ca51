#!/bin/bash
# Interleaved A/B timing of kernel builds on one GPU box (box-to-box noise is ~1 %, within a session ~0.1 %):
#   cp hmc.jl_amd/csrc/libhmcgibbs.so hmc.jl_amd/csrc/libhmcgibbs_<name>.so   for every build to compare, then
#   gpurun -- 'bash tools/ab.sh old new ...'      (reads them through HMCG_LIB; results in gpurun_out/ab.log)
mkdir -p gpurun_out; : > gpurun_out/ab.log
for i in 1 2 3; do
  for v in "$@"; do
    HMCG_LIB=libhmcgibbs_$v.so timeout -k 10 120 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extra 2>/dev/null | python -c "import sys,json; j=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$v', round(j['value']/1e6,3), round(j['roofline']['kernel_ms'],4))" >> gpurun_out/ab.log
  done
done
cat gpurun_out/ab.log
