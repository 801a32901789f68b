for cg in 16 8; do
  D3D_TILED_DSEG=128 D3D_TILED_STATS=1 D3D_TILED_CG=$cg python bench.py --steps 1 --warmup 0 --no-cpu-baseline 2>&1 | grep "d3d tiled" | head -2
done
