#!/bin/bash
# round 5, GPU call 4: fused location (tests + A/B)
cd "$(dirname "$0")/.." || exit 1
export HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/r5c5
mkdir -p $O
timeout 900 python -m pytest tests/test_layer_gpu.py -x -q -k "fused_location" > $O/pytest_fl.log 2>&1; echo "fl rc=$?"; tail -25 $O/pytest_fl.log
timeout 900 python tools/r5_headline_ab.py ${1:-l} > $O/ab.log 2>&1; echo "ab rc=$?"; tail -c 2500 $O/ab.log
cp gpurun_out/r5_headline_ab.json $O/ 2>/dev/null
