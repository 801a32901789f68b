#!/bin/bash
# GPU box: rocprofv3 kernel-trace stats of whole views (tools/model_bench.py, h16 mode, 2752 x 1856, 5 views) per model
#   tools/profile_views.sh r05 [models...]   -> gpurun_out/<round>_view_<model>_h16_kernel_stats.csv (+ the bench's own ms line)
R=${1:-r05}; shift
MODELS=${@:-"casmvsnet adamvs msrednet"}
out=$GRAFT_REPO_ROOT/gpurun_out/view_prof
rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
export D3D_CONV_PRECISION=h16
for m in $MODELS; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/$m -- python3 $GRAFT_REPO_ROOT/tools/model_bench.py --model $m --reps 3 > $out/$m.log 2>&1
  f=$(ls -t $out/$m/*/*kernel_stats.csv | head -1)
  cp $f $GRAFT_REPO_ROOT/gpurun_out/${R}_view_${m}_h16_kernel_stats.csv
  grep "ms per reference view" $out/$m.log
  head -12 $f | cut -c1-150
done
