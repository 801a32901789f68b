#!/bin/bash
# GPU box: an AdaMVS view in bf16 mode (tools/model_bench.py) with the round-4 fused kernels on / off, one box for all variants
cd "$GRAFT_REPO_ROOT" || exit 2
for f in "" head_fused gru_fused "gru_fused,head_fused" ""; do
  echo "== D3D_KERNELS_OFF=$f"
  D3D_KERNELS_OFF=$f D3D_CONV_PRECISION=bf16 python tools/model_bench.py --model adamvs --reps 5 2>&1 | grep -a "per reference view" | cut -c1-250
done
