#!/bin/bash
# round 5: TUTEL_OPT_GEMM_STORE -- bit identity of the three store policies, the GEMM tests on the rebuilt library, then the A/B
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/r5_store
mkdir -p $O
timeout 300 python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "store_policies or bit_identical or ep_layout or activations" > $O/pytest.log 2>&1; echo "tests rc=$?"; tail -4 $O/pytest.log
timeout 500 python tools/r5_store_ab.py > $O/ab.log 2>&1; echo "ab rc=$?"; tail -c 3000 $O/ab.log
