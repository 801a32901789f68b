#!/bin/bash
# One variant's numbers for the window kernel (CMD of tools/run_ab.sh with SRC=planesweep_window.hip EXP=1): per-workgroup
# statistics of the headline shape and of the cascade-stage shapes, then the timed runs.
export D3D_FORCE_PATH=window
python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-secondary 2>&1 | grep -a "window stats" | head -1
env -u D3D_WINDOW_STATS -u D3D_TILED_STATS python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('headline', d['value'], 'Mvox/s', d['roofline']['kernel_ms'], 'ms', d['roofline']['frac'])"
python tools/stage_sweep_bench.py window 2>&1 | grep -a "window stats" | awk 'NR%11==1'
env -u D3D_WINDOW_STATS -u D3D_TILED_STATS python tools/stage_sweep_bench.py window 2>&1 | grep -a "^stage"
