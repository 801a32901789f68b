for cg in 32 16; do
 for dbg in 1 2; do
  D3D_TILED_DEBUG=$dbg D3D_TILED_CG=$cg python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('DEBUG=$dbg CG=$cg', d['value'], 'Mvox/s', d['ms_per_step'], 'ms')"
 done
done
