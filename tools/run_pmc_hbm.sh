#!/bin/bash
# GPU box: HBM bytes per kernel of a python script -- FETCH_SIZE and WRITE_SIZE in passes of their own (never mixed with SQ
# counters: that combination aborted rocprofv3 on this pool) plus a kernel trace for the durations.
#   tools/run_pmc_hbm.sh <tag> <script.py> [args]      ->  gpurun_out/<tag>/hbm_summary.txt
tag=$1; shift
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
S="$GRAFT_REPO_ROOT/$1"; shift
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 $S "$@" > $out/trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/fetch -- python3 $S "$@" > $out/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/write -- python3 $S "$@" > $out/write.log 2>&1
python3 - <<PY
import csv, glob, collections
out = "$out"
per = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/fetch/*/*counter_collection.csv") + glob.glob(out + "/write/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        per[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
dur = {}
for f in glob.glob(out + "/trace/*/*kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        dur[r["Name"]] = (float(r["AverageNs"]), int(r["Calls"]))
with open(out + "/hbm_summary.txt", "w") as fo:
    fo.write("# per kernel (average over its calls): HBM read = 2 x FETCH_SIZE KiB (gfx950 half-reporting), HBM write = WRITE_SIZE KiB\n")
    for k in sorted(per, key=lambda k: -dur.get(k, (0, 0))[0] * dur.get(k, (0, 0))[1]):
        d = dur.get(k, (0.0, 0))
        rd = 2 * 1024 * sum(per[k].get("FETCH_SIZE", [0])) / max(1, len(per[k].get("FETCH_SIZE", [0])))
        wr = 1024 * sum(per[k].get("WRITE_SIZE", [0])) / max(1, len(per[k].get("WRITE_SIZE", [0])))
        if d[0] <= 0:
            continue
        fo.write("%-110s calls %4d avg %9.1f us  read %9.1f MB  write %9.1f MB  %7.1f GB/s\n" % (k[:110], d[1], d[0] / 1e3, rd / 1e6, wr / 1e6, (rd + wr) / d[0]))
print(open(out + "/hbm_summary.txt").read()[:5000])
PY
