#!/bin/bash
# Round measurement on the GPU box for the headline kernel: bench line, rocprofv3 kernel-trace stats, HBM traffic
# (separate FETCH_SIZE / WRITE_SIZE passes) and SQ counters.  Usage: tools/profile_round.sh r02
# Outputs under gpurun_out/<round>/ ; the summary (with the SHA-256 of the kernel's machine code it was taken from) is what gets
# copied to profiles/pmc_latest.json -- bench.py reports `roofline.traffic` only while the kernel's machine code in the built
# library hashes to what this profile recorded (_lib.kernel_code_sha256).  Make this the LAST GPU action of a round.
R=${1:-r02}
out=$GRAFT_REPO_ROOT/gpurun_out/$R
mkdir -p $out
cd $GRAFT_REPO_ROOT
python bench.py --steps 20 --warmup 5 > $out/bench.json 2> $out/bench.err
tail -1 $out/bench.json | cut -c1-300
cd /tmp && export TMPDIR=/tmp
B="python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-secondary"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- $B --steps 20 --warmup 5 > $out/trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch -- $B --steps 3 --warmup 1 > $out/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/pmc_write -- $B --steps 3 --warmup 1 > $out/pmc_write.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $out/pmc_sq -- $B --steps 3 --warmup 1 > $out/pmc_sq.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_INT32 SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $out/pmc_sq2 -- $B --steps 3 --warmup 1 > $out/pmc_sq2.log 2>&1
cp $out/trace/*/*kernel_stats.csv $out/kernel_stats.csv 2>/dev/null
python3 - <<PY
import csv, glob, collections, json, hashlib, sys
out = "$out"
root = "$GRAFT_REPO_ROOT"
sys.path.insert(0, root)
from deep3d_aerial_amd import _lib
PREFIX = "_ZN3d3d18sweep_tiled_kernelILi1ELi4ELi16EfLb0ELi4ELi4ELi2EEE"   # bench.HEADLINE_KERNEL_PREFIX
per = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/pmc_*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = "sweep" if "sweep_tiled" in r["Kernel_Name"] else "staging_copy" if "pack_channel_last" in r["Kernel_Name"] else None
        if k:
            per[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
            per[k]["_name"] = r["Kernel_Name"]
avg = {k: {c: sum(v) / len(v) for c, v in d.items() if c != "_name"} for k, d in per.items()}
res = {"round": "$R", "command": "python bench.py --steps 20 --warmup 5 (rocprofv3 passes: --steps 3 --warmup 1 --no-cpu-baseline --no-secondary); tools/profile_round.sh",
       "kernel": per["sweep"].get("_name"), "staging_copy_kernel": per["staging_copy"].get("_name"),
       "kernel_symbol_prefix": PREFIX, "kernel_code_sha256": _lib.kernel_code_sha256(PREFIX)}
for f in glob.glob(out + "/trace/*/*kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        if "sweep_tiled" in r["Name"]:
            res["kernel_trace_avg_ns"] = float(r["AverageNs"]); res["kernel_trace_calls"] = int(r["Calls"])
        if "pack_channel_last" in r["Name"]:
            res["staging_copy_trace_avg_ns"] = float(r["AverageNs"])
try:
    b = json.loads(open(out + "/bench.json").read().strip().splitlines()[-1])
    res["bench_launch_ms"] = b["roofline"]["kernel_ms"]; res["bench_frac"] = b["roofline"]["frac"]
    res["algorithmic_bytes_per_launch"] = b["roofline"]["algorithmic_bytes"]
except Exception as e:
    res["bench_error"] = repr(e)
fetch = {k: avg.get(k, {}).get("FETCH_SIZE") for k in ("sweep", "staging_copy")}
write = {k: avg.get(k, {}).get("WRITE_SIZE") for k in ("sweep", "staging_copy")}
res["FETCH_SIZE_KiB_reported"] = fetch
res["WRITE_SIZE_KiB_reported"] = write
res["fetch_correction"] = "x2: gfx950 FETCH_SIZE reports half the bytes (MI355X_MICROARCH.md, HBM; re-calibrated in round 1 with tools/fetch_calib.hip)"
if all(v is not None for v in list(fetch.values()) + list(write.values())):
    res["hbm_bytes_per_launch"] = int(sum(2 * 1024 * v for v in fetch.values()) + sum(1024 * v for v in write.values()))
res["sq_counters_per_launch"] = {c: v for c, v in avg.get("sweep", {}).items() if c not in ("FETCH_SIZE", "WRITE_SIZE")}
json.dump(res, open(out + "/pmc_latest.json", "w"), indent=1)
print(json.dumps(res)[:1800])
PY
