#!/bin/bash
# round 5, GPU call 3: scalar slot-map lookup in the ring kernels (tests + A/B)
cd "$(dirname "$0")/.." || exit 1
export HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/r5c4
mkdir -p $O
hipcc --offload-arch=gfx950 -O3 tools/scratch/sbuf_gather_probe.hip -o /tmp/sbuf_probe 2>/dev/null && /tmp/sbuf_probe
timeout 900 python -m pytest tests/test_ops_gpu.py -x -q -k "gemm or headline" > $O/pytest_ops.log 2>&1; echo "ops rc=$?"; tail -4 $O/pytest_ops.log
timeout 900 python -m pytest tests/test_layer_gpu.py -x -q -k "headline or dropless or fixture or graph" > $O/pytest_layer.log 2>&1; echo "layer rc=$?"; tail -4 $O/pytest_layer.log
timeout 900 python tools/r5_headline_ab.py ${1:-s} > $O/ab.log 2>&1; echo "ab rc=$?"; tail -c 3000 $O/ab.log
cp gpurun_out/r5_headline_ab.json $O/ 2>/dev/null
