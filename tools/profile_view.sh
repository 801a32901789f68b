#!/bin/bash
# GPU box: rocprofv3 kernel-trace stats of one CasMVSNet / AdaMVS view (tools/model_bench.py) -> gpurun_out/view_prof/
out=$GRAFT_REPO_ROOT/gpurun_out/view_prof
rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for m in casmvsnet adamvs; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/$m -- python3 $GRAFT_REPO_ROOT/tools/model_bench.py --model $m > $out/$m.log 2>&1
  f=$(ls -t $out/$m/*/*kernel_stats.csv | head -1)
  cp $f $out/${m}_kernel_stats.csv
  head -8 $out/${m}_kernel_stats.csv | cut -c1-160
done
