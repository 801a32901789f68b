#!/bin/bash
# CMD for tools/kvariants.sh gru_fused.hip: tools/gru2_debug.py under its three input modes (which term of h' carries the mismatch)
for m in "" nocand nostate nogate; do echo "mode=$m"; GRU2_MODE=$m python tools/gru2_debug.py 2>&1 | grep -a "run 0" | cut -c1-110; done
