#!/bin/bash
# GPU box: rebuild the library with the conv_stream phase counters and print the breakdown
set -e
mkdir -p gpurun_out
cd deep3d_aerial_amd/csrc
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-fast-math -ffp-contract=off -DD3D_CONV_STATS -c conv_stream.hip -o conv_stream.o
hipcc --offload-arch=gfx950 -shared -fPIC -o libdeep3d_planesweep.so planesweep.o planesweep_tiled.o regress.o conv.o conv_mfma.o conv_stream.o
cd ../..
timeout -k 10 300 python tools/conv_stats.py > gpurun_out/conv_stats.log 2>&1 || { tail -30 gpurun_out/conv_stats.log; exit 1; }
cat gpurun_out/conv_stats.log
