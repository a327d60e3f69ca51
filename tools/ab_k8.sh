#!/bin/bash
# A/B of the K=8 shape (configs[3]) between library builds, like tools/ab.sh: gpurun -- 'bash tools/ab_k8.sh old new'
mkdir -p gpurun_out; : > gpurun_out/ab_k8.log
for i in 1 2; do
  for v in "$@"; do
    HMCG_LIB=libhmcgibbs_$v.so timeout -k 10 200 python - <<PY >> gpurun_out/ab_k8.log
import bench, json
r = bench.shape_record("k8", 8, [5000] * 512, 1000, reps=2)
print("$v", round(r["value"]/1e6, 4), round(r["kernel_ms"], 2))
PY
  done
done
cat gpurun_out/ab_k8.log
