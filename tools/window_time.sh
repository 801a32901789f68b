#!/bin/bash
# CMD for tools/run_ab.sh SRC=planesweep_window.hip: production-build timings of a window-kernel variant on the cascade-stage
# shapes (TESTS=1: the sweep parity tests on the variant first)
[ -n "$TESTS" ] && python -m pytest tests/test_parity_gpu.py -x -q -k "window or sweep or variance or pair" 2>&1 | tail -2
export D3D_FORCE_PATH=window
env -u D3D_WINDOW_STATS -u D3D_TILED_STATS python tools/stage_sweep_bench.py window 2>&1 | grep -a "^stage"
