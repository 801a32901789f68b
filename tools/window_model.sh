#!/bin/bash
# CMD for tools/run_ab.sh SRC=planesweep_window.hip: the sweeps inside a CasMVSNet / AdaMVS forward (dispatcher's choice) and the views' times
python tools/sweep_in_model.py casmvsnet 2>&1 | awk '/path auto/{f=1} /path tiled/{f=0} f' | grep -a "ms$"
for m in casmvsnet adamvs; do D3D_CONV_PRECISION=bf16 python tools/model_bench.py --model $m --reps 5 2>&1 | grep -a "per reference view" | cut -c1-100; done
