#!/bin/bash
# CMD for tools/run_ab.sh (SRC=planesweep_window.hip): the stage shapes on the bench scene and inside a CasMVSNet view (model scene)
python tools/stage_sweep_bench.py auto 2>&1 | grep -a "^stage"
python tools/sweep_in_model.py casmvsnet 2>&1 | grep -a "ms$" | sed -n 4,6p
