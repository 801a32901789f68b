#!/bin/bash
# Counter passes (and a kernel trace) over a python script on the GPU box: tools/run_pmc_script.sh <tag> <script.py> [args]
# Per-kernel averages land in gpurun_out/<tag>/summary.txt
tag=$1; shift
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
S="$GRAFT_REPO_ROOT/$1"; shift
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 $S "$@" > $out/trace.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $out/pmc1 -- python3 $S "$@" > $out/pmc1.log 2>&1
rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_SALU SQ_WAIT_ANY --output-format csv -d $out/pmc2 -- python3 $S "$@" > $out/pmc2.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_DATA_FIFO_FULL SQ_INSTS_SMEM --output-format csv -d $out/pmc3 -- python3 $S "$@" > $out/pmc3.log 2>&1
python3 - <<PY
import csv, glob, collections
out = "$out"
per = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/pmc*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        per[r["Kernel_Name"][:110]][r["Counter_Name"]].append(float(r["Counter_Value"]))
dur = {}
for f in glob.glob(out + "/trace/*/*kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        dur[r["Name"][:110]] = (float(r["AverageNs"]), int(r["Calls"]))
with open(out + "/summary.txt", "w") as fo:
    for k in sorted(per, key=lambda k: -dur.get(k, (0, 0))[0] * dur.get(k, (0, 0))[1]):
        d = dur.get(k, (0, 0))
        fo.write("%s\n   avg %.1f us x %d calls\n" % (k, d[0] / 1e3, d[1]))
        for c, v in sorted(per[k].items()):
            fo.write("   %-32s %.4g\n" % (c, sum(v) / len(v)))
print(open(out + "/summary.txt").read()[:6000])
PY
