#!/bin/bash
# GPU box: run the cost-volume parity tests against the tiled kernel built with each given set of -D flags
cp deep3d_aerial_amd/csrc/libdeep3d_planesweep.so /tmp/keep.so
for flags in "$@"; do
  (cd deep3d_aerial_amd/csrc && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-fast-math -ffp-contract=off $flags -c -o /tmp/v.o planesweep_tiled.hip 2>/dev/null && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o libdeep3d_planesweep.so planesweep.o /tmp/v.o regress.o conv.o conv_mfma.o conv_stream.o) || { echo "build failed: $flags"; continue; }
  echo "[$flags]"; timeout -k 10 300 python -m pytest tests/test_parity_gpu.py -q -m gpu -k "aggregation or warp or variance or full_size" 2>&1 | tail -4 | cut -c1-200
done
cp /tmp/keep.so deep3d_aerial_amd/csrc/libdeep3d_planesweep.so
