#!/bin/bash
# GPU box: bench the tiled sweep kernel built with each given set of -D flags, e.g.
#   tools/run_ab.sh "-DRING_ALIGN=0" "-DRING_ALIGN=1"
cp deep3d_aerial_amd/csrc/libdeep3d_planesweep.so /tmp/keep.so
for flags in "$@"; do
  (cd deep3d_aerial_amd/csrc && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-fast-math -ffp-contract=off $flags -c -o /tmp/v.o planesweep_tiled.hip && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o libdeep3d_planesweep.so planesweep.o /tmp/v.o regress.o conv.o conv_mfma.o conv_stream.o fusion.o mapio.o) || { echo "build failed: $flags"; continue; }
  D3D_TILED_STATS=1 python bench.py --steps 1 --warmup 0 --no-cpu-baseline 2>&1 | grep "per-WG" | head -1 | cut -c1-240
  python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('[$flags]', d['value'], 'Mvox/s', d['ms_per_step'], 'ms', d['roofline']['frac'])"
done
cp /tmp/keep.so deep3d_aerial_amd/csrc/libdeep3d_planesweep.so
