#!/bin/bash
# GPU box: stall attribution counters of the sweep kernel (three --pmc passes, kernel-trace off), summary to
# gpurun_out/pmc2/summary.json.  Sums over XCDs / SEs as rocprofv3 reports them, averaged over the dispatches.
out=$GRAFT_REPO_ROOT/gpurun_out/pmc2
rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
run() { rocprofv3 --pmc $2 --output-format csv -d $out/$1 -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $out/$1.log 2>&1; }
run a "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_BUSY_CU_CYCLES"
run b "SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_INST_CYCLES_VMEM_WR"
run c "SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_LDS_LOAD SQ_INSTS_LDS_STORE SQ_IFETCH SQ_INST_LEVEL_LDS SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU_TRANS_F32"
run d "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_ADD_F32"
python3 - <<PY
import csv,glob,collections,json
out="$out"
res={}
for f in glob.glob(out+"/*/*/*counter_collection.csv"):
    tot=collections.defaultdict(float); cnt=collections.Counter()
    for r in csv.DictReader(open(f)):
        if 'sweep_tiled' in r['Kernel_Name']:
            tot[r['Counter_Name']]+=float(r['Counter_Value']); cnt[r['Counter_Name']]+=1
    for k in tot: res[k]=tot[k]/cnt[k]
json.dump(res, open(out+"/summary.json","w"), indent=1)
for k in sorted(res): print("%-34s %.4g" % (k, res[k]))
PY
