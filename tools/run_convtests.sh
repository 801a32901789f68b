#!/bin/bash
# GPU: conv-family parity tests, then the full GPU suite, then the model bench
set -e
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_parity_gpu.py -x -q -m gpu -k "conv or costreg or gru or pairnet" > gpurun_out/convtests.log 2>&1 || { tail -40 gpurun_out/convtests.log; exit 1; }
tail -3 gpurun_out/convtests.log
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/gputests.log 2>&1 || { tail -40 gpurun_out/gputests.log; exit 1; }
tail -3 gpurun_out/gputests.log
timeout -k 10 600 python tools/model_bench.py > gpurun_out/model_bench.log 2>&1 || { tail -40 gpurun_out/model_bench.log; exit 1; }
tail -30 gpurun_out/model_bench.log
