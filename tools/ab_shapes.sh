#!/bin/bash
# Interleaved A/B of library builds on other shapes than the headline (tools/ab.sh does that one): configs[3] (K=8, T=5000,
# 512 windows), the 2048-window and the production shapes, the signal path.  gpurun -- 'bash tools/ab_shapes.sh old new'
mkdir -p gpurun_out; : > gpurun_out/ab_shapes.log
for i in 1 2; do
  for v in "$@"; do
    HMCG_LIB=libhmcgibbs_$v.so timeout -k 10 300 python - <<PY >> gpurun_out/ab_shapes.log 2>/dev/null
import bench
r = bench.shape_record("k8", 8, [5000] * 512, 1000, reps=2)
a = bench.shape_record("w2048", 3, [1000] * 2048, 1000, reps=2)
b = bench.shape_record("prod460", 3, list(range(120, 580)), 1000, reps=3)
c = bench.shape_record("k4", 4, [1000] * 256, 500, reps=3)
print("$v", "k8 %.2f ms" % r["kernel_ms"], "w2048 %.3f ms" % a["kernel_ms"], "prod460 %.3f ms" % b["kernel_ms"], "k4 %.3f ms" % c["kernel_ms"])
PY
  done
done
cat gpurun_out/ab_shapes.log
