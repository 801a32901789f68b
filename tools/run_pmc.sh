# usage: run_pmc.sh <tag> [env...]  -- collects SQ counters for the bench kernel into gpurun_out/pmc_<tag>/
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
mkdir -p $GRAFT_REPO_ROOT/gpurun_out/pmc_$tag
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD" "SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM_WR SQ_INSTS_SMEM SQ_ACTIVE_INST_MISC SQ_INSTS_BRANCH SQ_ACTIVE_INST_VMEM SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_VMEM_TA_ADDR_FIFO_FULL" "GRBM_GUI_ACTIVE SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_THREAD_CYCLES_VALU SQ_CYCLES"; do
  n=$(echo $set | cut -d' ' -f1)
  env "$@" rocprofv3 --pmc $set --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_$tag/$n -- python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/pmc_$tag/$n.log 2>&1 || echo "pmc pass $n failed"
done
python3 - <<PY
import csv,glob,collections
tot=collections.defaultdict(float); cnt=collections.Counter()
for f in glob.glob("$GRAFT_REPO_ROOT/gpurun_out/pmc_$tag/*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if 'sweep' in r['Kernel_Name']:
            tot[r['Counter_Name']]+=float(r['Counter_Value']); cnt[r['Counter_Name']]+=1
for k in sorted(tot): print("%-24s %16.0f  (per launch, %d launches)"%(k, tot[k]/max(cnt[k],1), cnt[k]))
PY
