#!/bin/bash
# GPU box: what a ring-kernel build variant does to the cascade: stage shapes + the sweep kernels inside whole views.
# Use as CMD of tools/run_ab.sh, e.g.  NOEXP=1 CMD="tools/shallow_ab.sh" tools/run_ab.sh "-DD3D_SHALLOW_PLANES=0" "-DD3D_SHALLOW_PLANES=16"
python tools/stage_sweep_bench.py tiled 2>&1 | grep -a "^stage"
for m in casmvsnet adamvs; do
  D3D_CONV_PRECISION=bf16 python tools/model_bench.py --model $m --reps 3 2>&1 | grep -a "per reference\|sweep_tiled" | cut -c1-75,150-250
done
