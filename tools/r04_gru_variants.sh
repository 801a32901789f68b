#!/bin/bash
# GPU box: tools/gru_fused_bench.py on variants of csrc/gru_fused.hip built with the given -D flag sets (scratch libraries;
# -DD3D_GRU_PREFETCH=0|1, -DD3D_GRU_WAVES2=2 -- the timing-only -DD3D_GRU_X builds of the first half of round 4 are gone)
cd "$GRAFT_REPO_ROOT" || exit 2
CS=deep3d_aerial_amd/csrc
OBJS=$(make -s -C $CS print-objs)
VDIR=$(mktemp -d /tmp/d3d_gru.XXXXXX)
trap 'rm -rf "$VDIR"' EXIT
n=0
for flags in "$@"; do
  n=$((n+1)); V=$VDIR/v$n.so
  (cd $CS && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-fast-math -ffp-contract=off -w $flags -c -o $VDIR/v.o gru_fused.hip \
    && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o $V $(echo $OBJS | sed "s#gru_fused.o#$VDIR/v.o#")) || { echo "build failed: $flags"; continue; }
  echo "[$flags]"
  D3D_LIBRARY=$V python tools/gru_fused_bench.py 10 2>&1 | grep -av amdgpu.ids
done
