#!/bin/bash
# round 6: the evidence files quoted in DESIGN.md / csrc/expert_ffn.hip (run through gpurun from the repo root); outputs under gpurun_out/r06/
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r06
MODE=1 python tools/r6_ffn_probe.py > gpurun_out/r06/ffn_persistent_probe.json 2> gpurun_out/r06/ffn_probe.err
( echo "# tools/scratch/tile_bench.hip: the 128 x 256 ring tile at 64 experts x 128 rows, N = 2048, by K (weights per set: 0.54 GB at K = 2048, 2.1 GB at K = 8192; two sets alternate)"; for k in 2048 8192; do echo "## K = $k"; tools/scratch/tile_bench $k | grep "rep 2\|phases"; done ) > gpurun_out/r06/tile_bench.txt 2>&1
tools/r6_tie_probe2.sh > gpurun_out/r06/tie_replay_cost_by_tied_rows.txt 2>&1
for f in 1 0 1 0 1 0; do TUTEL_AMD_FFN_FUSED=$f python bench.py --steps 20 --warmup 5 --no_extra --no_cpu_baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('TUTEL_AMD_FFN_FUSED=$f ms_per_step', d['ms_per_step'], 'min', d['step_ms']['min'], 'median', d['step_ms']['median'], 'eager', d['launch_modes']['other']['ms_per_step'])"; done > gpurun_out/r06/ffn_fused_vs_two_launches_bench.txt
tools/r6_prof.sh ffnfused TUTEL_AMD_FFN_FUSED=1 > gpurun_out/r06/ffn_fused_kernel_stats_head.txt
cp gpurun_out/prof_ffnfused_kernel_stats.csv gpurun_out/r06/ffn_fused_bench_kernel_stats.csv
