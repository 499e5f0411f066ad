#!/bin/bash
# round 5, last call: write-through peer rows (fc2 epilogue of the IPC transport) and encode rows -- the IPC tests with the rank processes
# sharing the GPU and the encode tests; only if they pass: re-stamp profiles (expert_gemm.hip changed) and the bench line
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/r5_peer_store
mkdir -p $O
timeout 200 python -m pytest tests/test_ep_ipc_one_gpu.py tests/test_ops_gpu.py -x -q -m gpu -k "ipc_transport or encode or multi_gpu_stress or store_policies" > $O/pytest.log 2>&1
rc=$?; echo "tests rc=$rc"; tail -3 $O/pytest.log
[ $rc -ne 0 ] && exit 1
GIT_HEAD=$1 bash tools/profile_r05.sh quick > $O/profile.log 2>&1; tail -2 $O/profile.log
