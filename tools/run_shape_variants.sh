#!/bin/bash
# GPU box: parity + bench of the tiled kernel with alternative tile shapes.  usage: "TH NPIXW NSUB" ...
cp deep3d_aerial_amd/csrc/libdeep3d_planesweep.so /tmp/keep.so
for spec in "$@"; do
  set -- $spec; th=$1; npw=$2; nsub=$3
  sed -e "s/constexpr int TH = [0-9]*;/constexpr int TH = $th;/" -e "s/constexpr int NPIXW = [0-9]*;/constexpr int NPIXW = $npw;/" -e "s/constexpr int NSUB = [0-9]*;/constexpr int NSUB = $nsub;/" deep3d_aerial_amd/csrc/planesweep_tiled.hip > deep3d_aerial_amd/csrc/planesweep_tiled_v.hip
  (cd deep3d_aerial_amd/csrc && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-fast-math -ffp-contract=off -c -o /tmp/v.o planesweep_tiled_v.hip 2>/dev/null && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o libdeep3d_planesweep.so planesweep.o /tmp/v.o regress.o conv.o conv_mfma.o conv_stream.o) || { echo "build failed: $spec"; rm -f deep3d_aerial_amd/csrc/planesweep_tiled_v.hip; continue; }
  rm -f deep3d_aerial_amd/csrc/planesweep_tiled_v.hip
  echo "[TH=$th NPIXW=$npw NSUB=$nsub]"
  timeout -k 10 300 python -m pytest tests/test_parity_gpu.py -x -q -m gpu -k "aggregation_vs_oracle and tiled" 2>&1 | tail -1
  D3D_TILED_STATS=1 python bench.py --steps 1 --warmup 0 --no-cpu-baseline 2>&1 | grep "d3d tiled" | head -2
  python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], 'Mvox/s', d['ms_per_step'], 'ms', d['roofline']['frac'])"
done
cp /tmp/keep.so deep3d_aerial_amd/csrc/libdeep3d_planesweep.so
