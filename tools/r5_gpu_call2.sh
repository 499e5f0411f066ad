#!/bin/bash
# round 5, GPU call 2: split-K gate projection (tests + A/B), cache warm-up A/B
cd "$(dirname "$0")/.." || exit 1
export HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/r5c2
mkdir -p $O
timeout 600 python -m pytest tests/test_ops_gpu.py -x -q -k "gate_proj or gate_topk or fused_softmax" > $O/pytest_ops.log 2>&1; echo "ops rc=$?"; tail -5 $O/pytest_ops.log
timeout 900 python -m pytest tests/test_layer_gpu.py -x -q -k "gate_projection_inside or low_precision" > $O/pytest_layer.log 2>&1; echo "layer rc=$?"; tail -8 $O/pytest_layer.log
timeout 900 python tools/r5_headline_ab.py ${1:-gfw} > $O/ab.log 2>&1; echo "ab rc=$?"; tail -c 6000 $O/ab.log
cp gpurun_out/r5_headline_ab.json $O/ 2>/dev/null
