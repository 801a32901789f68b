#!/bin/bash
# CMD for `SRC=conv2d_zs.hip tools/run_ab.sh "" "-DD3D_Z2_TPER_MODEL=0"`: one FeatureNet forward (total device time of the profiler
# table) and the per-stage view times of the three models on a variant of the 2-D tile kernels.
for m in casmvsnet adamvs; do
  python tools/feature_bench.py $m 2>/dev/null | grep -a "Self CUDA time total" | sed "s/^/feature net $m: /"
done
for m in casmvsnet adamvs msrednet; do python tools/stage_times.py $m "" 2>&1 | grep -a view; done
