#!/bin/bash
# GPU box: an AdaMVS view in bf16 mode with the fused conv-GRU cell on / off (tools/model_bench.py), key kernels only
cd "$GRAFT_REPO_ROOT" || exit 2
for f in "" gru_fused; do
  echo "== D3D_KERNELS_OFF=$f"
  D3D_KERNELS_OFF=$f D3D_CONV_PRECISION=bf16 python tools/model_bench.py --model adamvs --reps 3 2>&1 | grep -a "per reference view\|gru_cell\|conv2d_zs\|convt2d\|conv2d_s2\|online_regress\|Self CUDA time" | cut -c1-250
done
