#!/bin/bash
# Round 4: SQ / LDS counters of the two arms of the 128-row regime at the headline shape (one counter group per rocprofv3 run, no trace
# domains beside --pmc):  bash tools/r4_pmc_arms.sh   -> gpurun_out/r4_pmc/{impl1,impl4}/summary.txt
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r4_pmc
TUTEL_AMD_GEMM_IMPL=1 python tools/pmc_run.py gpurun_out/r4_pmc/impl1 expert_gemm -- python tools/gemm_pmc_probe.py 64 128 2048 2048 > gpurun_out/r4_pmc/impl1.txt 2>&1
TUTEL_AMD_GEMM_IMPL=4 python tools/pmc_run.py gpurun_out/r4_pmc/impl4 expert_gemm -- python tools/gemm_pmc_probe.py 64 128 2048 2048 > gpurun_out/r4_pmc/impl4.txt 2>&1
tail -20 gpurun_out/r4_pmc/impl1.txt gpurun_out/r4_pmc/impl4.txt
find gpurun_out/r4_pmc -name "*.csv" -delete; find gpurun_out/r4_pmc -type d -empty -delete
