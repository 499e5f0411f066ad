#!/bin/bash
# round 5, last GPU call of the round: the bench line on the final tree (cpu_baseline now times the reference's compiled kernels) and the
# SQ / LDS counter passes of this round's kernels (one counter group per rocprofv3 run, no trace domains beside --pmc):
#   fc1 / fc2 inside the bench (fc1 = the fused-location form of the 128 x 256 ring), and the two MFMA-bound launches of an 8-way
#   expert-parallel rank (8 x 1024 x 2048 x 2048 on the ping-pong kernel, its 4-expert pipeline stage on the 256 x 128 ring).
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/r5_final
mkdir -p $O
timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench_line.json 2> $O/bench_line.err; echo "bench rc=$?"
timeout 400 python tools/pmc_run.py $O/pmc_bench expert_gemm -- python bench.py --eager --steps 20 --warmup 5 --settle 20 --no_cpu_baseline --no_extra > $O/pmc_bench.txt 2>&1
timeout 200 python tools/pmc_run.py $O/pmc_rank expert_gemm -- python tools/gemm_pmc_probe.py 8 1024 2048 2048 > $O/pmc_rank.txt 2>&1
timeout 200 python tools/pmc_run.py $O/pmc_stage expert_gemm -- python tools/gemm_pmc_probe.py 4 1024 2048 2048 > $O/pmc_stage.txt 2>&1
find $O -name "*.csv" -delete; find $O -name "*.db" -delete; find $O -type d -empty -delete
tail -c 600 $O/bench_line.json; tail -5 $O/pmc_bench.txt $O/pmc_rank.txt $O/pmc_stage.txt
