set -e
timeout -k 10 600 python -m pytest tests/test_parity_gpu.py -m gpu -q -x -k "not model" > gpurun_out/pytest_gpu.log 2>&1 || { tail -30 gpurun_out/pytest_gpu.log | cut -c1-250; exit 1; }
tail -1 gpurun_out/pytest_gpu.log
for dseg in ${DSEGS:-128}; do
for cg in ${CGS:-16 8}; do
  export D3D_TILED_DSEG=$dseg
  D3D_TILED_STATS=1 D3D_TILED_CG=$cg python bench.py --steps 1 --warmup 0 --no-cpu-baseline 2>&1 | grep "d3d tiled" | head -2
  D3D_TILED_CG=$cg python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('DSEG=$dseg CG=$cg', d['value'], 'Mvox/s', d['ms_per_step'], 'ms', d['roofline']['frac'])"
done
done
