#!/bin/bash
# GPU box: dynamic instruction counts per wave of the window kernel's phases -- -DD3D_EXPERIMENTS variants with the staging and / or
# the sweep left out (results wrong), one rocprofv3 counter pass each over tools/stage_sweep_case.py <stage>:
#   tools/window_insts.sh stage3 ["extra -D flags"]
[ -n "$GRAFT_REPO_ROOT" ] || { echo "GRAFT_REPO_ROOT is not set"; exit 2; }
cd "$GRAFT_REPO_ROOT" || exit 2
STAGE=${1:-stage3}; EXTRA=$2
CS=deep3d_aerial_amd/csrc
OBJS=$(make -s -C $CS print-objs)
VDIR=$(mktemp -d /tmp/d3d_wi.XXXXXX)
trap 'rm -rf "$VDIR"' EXIT
n=0
for flags in "" "-DD3D_WX_NOSTAGE" "-DD3D_WX_NOSWEEP" "-DD3D_WX_NOSTAGE -DD3D_WX_NOSWEEP"; do
  n=$((n+1)); V=$VDIR/v$n.so
  (cd $CS && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-fast-math -ffp-contract=off -w -DD3D_EXPERIMENTS $EXTRA $flags -c -o $VDIR/v.o planesweep_window.hip \
    && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o $V $(echo $OBJS | sed "s#planesweep_window.o#$VDIR/v.o#")) || { echo "build failed: $flags"; continue; }
  export D3D_LIBRARY=$V
  O=$GRAFT_REPO_ROOT/gpurun_out/wi_$n; rm -rf $O; mkdir -p $O
  (cd /tmp && TMPDIR=/tmp rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES --output-format csv -d $O -- python3 $GRAFT_REPO_ROOT/tools/stage_sweep_case.py $STAGE 2 > $O/log.txt 2>&1) || echo "pmc pass failed"
  python3 - <<PY
import csv,glob,collections
per=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$O/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if 'sweep_window' in r['Kernel_Name']:
            per[r['Kernel_Name'].split('<')[1].split('>')[0]][r['Counter_Name']].append(float(r['Counter_Value']))
print("[$STAGE $EXTRA $flags]")
for k in sorted(per):
    c={n:sum(v)/len(v) for n,v in per[k].items()}
    w=c.get('SQ_WAVES',1)
    print("  <%s> per wave: VALU %.0f SALU %.0f LDS %.0f SMEM %.0f VMEM_RD %.0f VMEM_WR %.0f wave cycles %.0f" % (k, c.get('SQ_INSTS_VALU',0)/w, c.get('SQ_INSTS_SALU',0)/w, c.get('SQ_INSTS_LDS',0)/w, c.get('SQ_INSTS_SMEM',0)/w, c.get('SQ_INSTS_VMEM_RD',0)/w, c.get('SQ_INSTS_VMEM_WR',0)/w, 4*c.get('SQ_WAVE_CYCLES',0)/w))
PY
  unset D3D_LIBRARY
done
