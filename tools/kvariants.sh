#!/bin/bash
# GPU box: run a command on variants of ONE kernel source built with the given -D flag sets (scratch libraries, D3D_LIBRARY):
#   tools/kvariants.sh conv_t2p.hip "python tools/t2p_bench.py" "" "-DD3D_T2P_NZ=2048" ...
cd "$GRAFT_REPO_ROOT" || exit 2
SRC=$1; CMD=$2; shift 2
CS=deep3d_aerial_amd/csrc
OBJS=$(make -s -C $CS print-objs)
VDIR=$(mktemp -d /tmp/d3d_var.XXXXXX)
trap 'rm -rf "$VDIR"' EXIT
n=0
for flags in "$@"; do
  n=$((n+1)); V=$VDIR/v$n.so
  (cd $CS && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-fast-math -ffp-contract=off -w $flags -c -o $VDIR/v.o $SRC \
    && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o $V $(echo $OBJS | sed "s#${SRC%.hip}.o#$VDIR/v.o#")) || { echo "build failed: $flags"; continue; }
  echo "[$flags]"
  D3D_LIBRARY=$V $CMD 2>&1 | grep -av amdgpu.ids
done
