#!/bin/bash
# GPU box: kernel-trace stats and MFMA counters of the regulariser conv kernels -> gpurun_out/conv_prof/
out=$GRAFT_REPO_ROOT/gpurun_out/conv_prof
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 $GRAFT_REPO_ROOT/tools/conv_prof_case.py > $out/stats.log 2>&1 || { tail -20 $out/stats.log; exit 1; }
for set in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVES" "SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAVE_CYCLES" "SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT"; do
  n=$(echo $set | cut -d' ' -f1)
  rocprofv3 --pmc $set --output-format csv -d $out/$n -- python3 $GRAFT_REPO_ROOT/tools/conv_prof_case.py > $out/$n.log 2>&1 || echo "pmc pass $n failed"
done
python3 - <<PY
import csv, glob, collections
out = "$out"
dur = collections.defaultdict(list)
for f in glob.glob(out + "/stats/*/*kernel_trace.csv"):
    for r in csv.DictReader(open(f)):
        if "conv_" in r["Kernel_Name"]:
            dur[r["Kernel_Name"]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
tot = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(collections.Counter)
for f in glob.glob(out + "/SQ_*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "conv_" in r["Kernel_Name"]:
            tot[r["Kernel_Name"]][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[r["Kernel_Name"]][r["Counter_Name"]] += 1
lines = ["# rocprofv3 on tools/conv_prof_case.py (stage-3 CostRegNet layers at 2752x1856, fp32 MFMA 16x16x4), MI355X",
         "# MFMA issue-slot share = SQ_INSTS_MFMA * 32 cycles / (kernel duration * 2.4 GHz * 1024 SIMDs); counters are per launch;",
         "# launch order in the case file: conv0 8->8, prob 8->1, conv2 16->16, conv11 transposed 16->8"]
for k in sorted(dur):
    short = k.replace("void d3d::(anonymous namespace)::", "").split("(")[0]
    d = sum(dur[k][1:]) / max(len(dur[k]) - 1, 1) if len(dur[k]) > 1 else dur[k][0]
    c = {n: tot[k][n] / max(cnt[k][n], 1) for n in tot[k]}
    line = "%-34s avg %8.1f us over %d launches" % (short, d, len(dur[k]))
    if "SQ_INSTS_MFMA" in c:
        line += " | SQ_INSTS_MFMA %.3e -> %.1f %% of MFMA issue slots" % (c["SQ_INSTS_MFMA"], 100 * c["SQ_INSTS_MFMA"] * 32 / (d * 1e-6 * 2.4e9 * 1024))
    lines.append(line)
    lines.append("    " + "  ".join("%s=%.4g" % (n, c[n]) for n in sorted(c)))
open(out + "/summary.txt", "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
PY
