#!/bin/bash
# GPU box: shallow (6-wave, half-LDS) vs default workgroups of the ring kernel at the cascade-stage shapes (experiments build)
for s in 0 1; do
  echo "== D3D_TILED_SHALLOW=$s"
  D3D_TILED_SHALLOW=$s python tools/stage_sweep_bench.py tiled 2>&1 | grep -a "^stage" 
  D3D_TILED_STATS=1 D3D_TILED_SHALLOW=$s python tools/stage_sweep_bench.py tiled 2>&1 | grep -a "tiled stats\|per-WG" | sort | uniq -c | sort -rn | awk '{ $1=""; print }' | cut -c1-330 | grep -a "CH=8" | head -4
done
