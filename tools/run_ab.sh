#!/bin/bash
# GPU box: bench the tiled sweep kernel built with each given set of -D flags, e.g.
#   tools/run_ab.sh "" "-DD3D_NOSTORE" "-DD3D_LDS_PIPE=2"
# Each variant is linked to a scratch library of its own and selected through D3D_LIBRARY (deep3d_aerial_amd/_lib.py):
# the in-tree production library is never touched.  EXP=1 builds the variants with -DD3D_EXPERIMENTS (cycle statistics,
# D3D_TILED_* switches; prints the per-workgroup statistics).  CMD="..." runs that command instead of bench.py;
# C5=f16|f32 runs tools/config5_bench.py.  PMC=1 adds one rocprofv3 counter pass per variant.  SRC=planesweep_window.hip builds
# variants of the window kernel instead (D3D_WINDOW_STATS=1 is set beside D3D_TILED_STATS=1 for CMD runs).
[ -n "$GRAFT_REPO_ROOT" ] || { echo "GRAFT_REPO_ROOT is not set"; exit 2; }
cd "$GRAFT_REPO_ROOT" || exit 2
CS=deep3d_aerial_amd/csrc
OBJS=$(make -s -C $CS print-objs)
VDIR=$(mktemp -d /tmp/d3d_ab.XXXXXX)
trap 'rm -rf "$VDIR"' EXIT
EXPF=; [ -n "$EXP" ] && EXPF=-DD3D_EXPERIMENTS
SRC=${SRC:-planesweep_tiled.hip}
n=0
for flags in "$@"; do
  n=$((n+1)); V=$VDIR/v$n.so
  (cd $CS && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-fast-math -ffp-contract=off -w $EXPF $flags -c -o $VDIR/v.o $SRC \
    && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o $V $(echo $OBJS | sed "s#${SRC%.hip}.o#$VDIR/v.o#")) || { echo "build failed: $flags"; continue; }
  export D3D_LIBRARY=$V
  if [ -n "$CMD" ]; then
    echo "[$flags]"; D3D_TILED_STATS=1 D3D_WINDOW_STATS=1 $CMD 2>&1 | grep -av amdgpu.ids | cut -c1-400
  elif [ -n "$C5" ]; then
    D3D_TILED_STATS=1 D3D_FORCE_PATH=tiled python tools/config5_bench.py $C5 tiled 2>&1 | grep -a "per-WG\|tiled stats\|config 5" | tail -3 | cut -c1-330
  else
    [ -n "$EXP" ] && D3D_TILED_STATS=1 python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-secondary 2>&1 | grep -a "per-WG\|tiled stats" | head -2 | cut -c1-330
    python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('[$flags]', d['value'], 'Mvox/s', d['ms_per_step'], 'ms', d['roofline']['frac'])"
  fi
  if [ -n "$PMC" ]; then
    O=$GRAFT_REPO_ROOT/gpurun_out/pmc_ab; rm -rf $O; mkdir -p $O
    (cd /tmp && TMPDIR=/tmp rocprofv3 --pmc ${PMCSET:-SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_BUSY_CU_CYCLES SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAVE_CYCLES} --output-format csv -d $O -- python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-secondary > $O/log.txt 2>&1) || echo "pmc pass failed"
    python3 - <<PY
import csv,glob,collections
tot=collections.defaultdict(float); cnt=collections.Counter()
for f in glob.glob("$O/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if 'sweep_tiled' in r['Kernel_Name']:
            tot[r['Counter_Name']]+=float(r['Counter_Value']); cnt[r['Counter_Name']]+=1
print("  pmc:", "  ".join("%s %.4g"%(k, tot[k]/max(cnt[k],1)) for k in sorted(tot)))
PY
  fi
  unset D3D_LIBRARY
done
