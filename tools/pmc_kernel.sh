#!/bin/bash
# GPU box: SQ / LDS counters of ONE kernel of a command (separate --pmc passes, no trace domains), averaged per dispatch.
# usage: tools/pmc_kernel.sh OUTDIR 'KERNEL_NAME_SUBSTRING' -- PROGRAM ARGS...   (PROGRAM is python3 or a binary: no wrappers)
usage() { echo "usage: tools/pmc_kernel.sh OUTDIR KERNEL_NAME_SUBSTRING -- PROGRAM ARGS..."; }
[ -n "$GRAFT_REPO_ROOT" ] && [ -n "$1" ] && [ -n "$2" ] && [ "$3" = "--" ] && [ $# -ge 4 ] || { usage; exit 2; }
out=$(realpath -m "$GRAFT_REPO_ROOT/$1"); root=$(realpath -m "$GRAFT_REPO_ROOT")
# OUTDIR is deleted and recreated: it has to be a directory strictly INSIDE the checkout's gpurun_out/
case "$out" in "$root"/gpurun_out/?*) ;; *) echo "OUTDIR must lie under gpurun_out/ (got $out)"; exit 2;; esac
pat=$2; shift 3
rm -rf "$out"; mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
run() { rocprofv3 --pmc $2 --output-format csv -d "$out/$1" -- "${@:3}" > "$out/$1.log" 2>&1 || echo "pass $1 failed"; }
run a "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_BUSY_CU_CYCLES" "$@"
run b "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "$@"
run c "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_LDS_LOAD SQ_INSTS_LDS_STORE" "$@"
run d "GRBM_GUI_ACTIVE SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INST_CYCLES_VMEM_WR SQ_THREAD_CYCLES_VALU" "$@"
python3 - "$out" "$pat" <<'PY'
import csv,glob,collections,json,sys
out,pat=sys.argv[1],sys.argv[2]
res={}
for f in glob.glob(out+"/*/*/*counter_collection.csv"):
    tot=collections.defaultdict(float); cnt=collections.Counter()
    for r in csv.DictReader(open(f)):
        if pat in r['Kernel_Name']:
            tot[r['Counter_Name']]+=float(r['Counter_Value']); cnt[r['Counter_Name']]+=1
    for k in tot: res[k]=tot[k]/cnt[k]; res.setdefault('_dispatches',{})[k]=cnt[k]
json.dump(res, open(out+"/summary.json","w"), indent=1)
for k in sorted(k for k in res if k[0]!='_'): print("%-34s %.5g" % (k, res[k]))
w=res.get('SQ_WAVE_CYCLES')
if w:
    print("WAIT_ANY/WAVE_CYCLES %.3f  ACTIVE_INST_ANY/WAVE_CYCLES %.3f" % (res.get('SQ_WAIT_ANY',0)/w, res.get('SQ_ACTIVE_INST_ANY',0)/w))
if res.get('SQ_LDS_IDX_ACTIVE'):
    print("LDS conflict share %.3f" % (res.get('SQ_LDS_BANK_CONFLICT',0)/res['SQ_LDS_IDX_ACTIVE']))
PY
