#!/bin/bash
# Runs GPU steps one after the other on the GPU box, each under its own `timeout -k 10`; a step that fails with an
# ordinary error does not stop the sequence, a step that is KILLED (timeout / signal) does: nothing is started on a GPU
# that may be hung.  usage: tools/gpu_seq.sh OUTDIR  "SECONDS|name|command" ...
out=$1; shift
mkdir -p "$out"
for spec in "$@"; do
    t=${spec%%|*}; rest=${spec#*|}; name=${rest%%|*}; cmd=${rest#*|}
    echo "== $name (limit ${t}s): $cmd"
    timeout -k 10 "$t" bash -c "$cmd" > "$out/$name.log" 2>&1
    rc=$?
    echo "== $name rc=$rc"; tail -n 6 "$out/$name.log"
    if [ $rc -eq 124 ] || [ $rc -ge 128 ]; then echo "== $name was killed: stopping"; exit $rc; fi
done
exit 0
