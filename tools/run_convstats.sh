#!/bin/bash
# GPU box: rebuild the library with the conv_stream phase counters (-DD3D_CONV_STATS), print the breakdown, and restore
# the clean library.  The instrumented object goes to /tmp and the link line comes from the Makefile's object list, so
# nothing instrumented is left in the tree.
CS=deep3d_aerial_amd/csrc
mkdir -p gpurun_out
cp $CS/libdeep3d_planesweep.so /tmp/keep.so
trap 'cp /tmp/keep.so $GRAFT_REPO_ROOT/'$CS'/libdeep3d_planesweep.so' EXIT
OBJS=$(make -s -C $CS print-objs)
(cd $CS && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-fast-math -ffp-contract=off -w -DD3D_EXPERIMENTS -DD3D_CONV_STATS -c -o /tmp/cs.o conv_stream.hip \
  && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o libdeep3d_planesweep.so $(echo $OBJS | sed 's#conv_stream.o#/tmp/cs.o#')) || { echo "build failed"; exit 1; }
timeout -k 10 300 python tools/conv_stats.py > gpurun_out/conv_stats.log 2>&1 || { tail -30 gpurun_out/conv_stats.log; exit 1; }
cat gpurun_out/conv_stats.log
