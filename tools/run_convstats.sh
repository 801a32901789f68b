#!/bin/bash
# GPU box: link a copy of the library with the conv_stream phase counters (-DD3D_CONV_STATS) to a scratch path, select it
# through D3D_LIBRARY (deep3d_aerial_amd/_lib.py) and print the breakdown.  The in-tree library is never touched.
[ -n "$GRAFT_REPO_ROOT" ] || { echo "GRAFT_REPO_ROOT is not set"; exit 2; }
cd "$GRAFT_REPO_ROOT" || exit 2
CS=deep3d_aerial_amd/csrc
mkdir -p gpurun_out
VDIR=$(mktemp -d /tmp/d3d_cs.XXXXXX)
trap 'rm -rf "$VDIR"' EXIT
OBJS=$(make -s -C $CS print-objs)
(cd $CS && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-fast-math -ffp-contract=off -w -DD3D_EXPERIMENTS -DD3D_CONV_STATS -c -o $VDIR/cs.o conv_stream.hip \
  && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o $VDIR/lib.so $(echo $OBJS | sed "s#conv_stream.o#$VDIR/cs.o#")) || { echo "build failed"; exit 1; }
D3D_LIBRARY=$VDIR/lib.so timeout -k 10 300 python tools/conv_stats.py > gpurun_out/conv_stats.log 2>&1 || { tail -30 gpurun_out/conv_stats.log; exit 1; }
cat gpurun_out/conv_stats.log
