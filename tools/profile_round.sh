#!/bin/bash
# Round measurement on the GPU box: bench line, rocprofv3 kernel-trace stats, HBM traffic counters.
# Outputs under gpurun_out/r01/ (copied into profiles/ afterwards).
set -x
out=$GRAFT_REPO_ROOT/gpurun_out/r01
mkdir -p $out
cd $GRAFT_REPO_ROOT
python bench.py --steps 20 --warmup 5 > $out/bench.json 2> $out/bench.err
tail -1 $out/bench.json | cut -c1-400
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $out/trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $out/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/pmc_write -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $out/pmc_write.log 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $out/pmc_tcc -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $out/pmc_tcc.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $out/pmc_sq -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $out/pmc_sq.log 2>&1
python3 - <<PY
import csv,glob,collections,json,os
out="$out"
res={}
for f in glob.glob(out+"/pmc_*/*/*counter_collection.csv"):
    tot=collections.defaultdict(float); cnt=collections.Counter()
    for r in csv.DictReader(open(f)):
        if 'sweep_tiled' in r['Kernel_Name']:
            tot[r['Counter_Name']]+=float(r['Counter_Value']); cnt[r['Counter_Name']]+=1
    for k in tot: res[k]=tot[k]/cnt[k]
for f in glob.glob(out+"/trace/*/*kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        if 'sweep_tiled' in r['Name']:
            res['kernel_stats']={k:r[k] for k in r}
        if 'pack_channel_last' in r['Name']:
            res['pack_kernel_stats']={k:r[k] for k in r}
json.dump(res, open(out+"/summary.json","w"), indent=1)
print(json.dumps(res)[:1500])
PY
