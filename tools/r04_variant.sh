#!/bin/bash
# GPU box: build ONE variant of the ring kernel (flags = $1) into a scratch library and run bench.py on it WITH the oracle
# parity leg (bench.py --no-secondary), printing time, roofline fraction and parity; EXP=1 adds the per-workgroup statistics.
#   tools/r04_variant.sh "-DD3D_DEV_ONLY_HEADLINE -DD3D_DEV_DEEP32" [more flag sets ...]
[ -n "$GRAFT_REPO_ROOT" ] || { echo "GRAFT_REPO_ROOT is not set"; exit 2; }
cd "$GRAFT_REPO_ROOT" || exit 2
CS=deep3d_aerial_amd/csrc
OBJS=$(make -s -C $CS print-objs)
VDIR=$(mktemp -d /tmp/d3d_var.XXXXXX)
trap 'rm -rf "$VDIR"' EXIT
n=0
for flags in "$@"; do
  n=$((n+1)); V=$VDIR/v$n.so
  (cd $CS && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-fast-math -ffp-contract=off -w -DD3D_EXPERIMENTS $flags -c -o $VDIR/v.o planesweep_tiled.hip \
    && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o $V $(echo $OBJS | sed "s#planesweep_tiled.o#$VDIR/v.o#")) || { echo "build failed: $flags"; continue; }
  export D3D_LIBRARY=$V
  echo "[$flags]"
  D3D_TILED_STATS=1 python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-secondary 2>&1 | grep -a "per-WG\|tiled stats" | head -2 | cut -c1-330
  python bench.py --steps 10 --warmup 3 --no-secondary 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('  ', d['value'], 'Mvox/s', d['ms_per_step'], 'ms frac', d['roofline']['frac'], '| parity', d.get('parity',{}).get('variance_rel_l1'), d.get('parity',{}).get('ok'))"
  unset D3D_LIBRARY
done
