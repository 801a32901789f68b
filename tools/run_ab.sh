#!/bin/bash
# GPU box: bench the tiled sweep kernel built with each given set of -D flags, e.g.
#   tools/run_ab.sh "-DRING_ALIGN=0" "-DRING_ALIGN=1"
# Variants are built with -DD3D_EXPERIMENTS (the D3D_TILED_* switches and the cycle statistics exist only there).
# C5=f16|f32 runs tools/config5_bench.py instead of bench.py; CMD="..." runs that command instead (statistics on).  PMC=1 adds one rocprofv3 counter pass (LDS conflicts / activity, VALU instructions) per variant.
# The variant library is linked from the Makefile's object list and the clean library is restored on exit.
CS=deep3d_aerial_amd/csrc
cp $CS/libdeep3d_planesweep.so /tmp/keep.so
trap 'cp /tmp/keep.so $GRAFT_REPO_ROOT/'$CS'/libdeep3d_planesweep.so' EXIT
OBJS=$(make -s -C $CS print-objs)
EXP=-DD3D_EXPERIMENTS
[ -n "$NOEXP" ] && EXP=   # NOEXP=1: the production build of the variant (no cycle counters in the kernel)
for flags in "$@"; do
  (cd $CS && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-fast-math -ffp-contract=off -w $EXP $flags -c -o /tmp/v.o planesweep_tiled.hip \
    && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o libdeep3d_planesweep.so $(echo $OBJS | sed 's#planesweep_tiled.o#/tmp/v.o#')) || { echo "build failed: $flags"; continue; }
  if [ -n "$CMD" ]; then   # any other command (e.g. CMD="python tools/stage_sweep_bench.py tiled"), with the cycle statistics on
    echo "[$flags]"; D3D_TILED_STATS=1 $CMD 2>&1 | grep -av amdgpu.ids | cut -c1-400
  elif [ -n "$C5" ]; then   # BASELINE config 5 shape instead of the bench (C5=f16|f32)
    D3D_TILED_STATS=1 D3D_FORCE_PATH=tiled python tools/config5_bench.py $C5 tiled 2>&1 | grep -a "per-WG\|tiled stats\|config 5" | tail -3 | cut -c1-330
  else
  D3D_TILED_STATS=1 python bench.py --steps 1 --warmup 0 --no-cpu-baseline 2>&1 | grep -a "per-WG\|tiled stats" | head -2 | cut -c1-330
  python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('[$flags]', d['value'], 'Mvox/s', d['ms_per_step'], 'ms', d['roofline']['frac'])"
  fi
  if [ -n "$PMC" ]; then
    R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmc_ab; rm -rf $O; mkdir -p $O
    (cd /tmp && TMPDIR=/tmp rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAVE_CYCLES --output-format csv -d $O -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/log.txt 2>&1) || echo "pmc pass failed"
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
done
