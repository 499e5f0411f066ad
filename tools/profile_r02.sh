#!/bin/bash
# Round-2 profile collection on the GPU box (run through gpurun from the repo root):
#   bash tools/profile_r02.sh
# Writes rocprofv3 summaries under gpurun_out/prof_r02/; the files worth keeping are copied to profiles/ by hand.
set -x
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_r02
mkdir -p $OUT
# 1. kernel trace + stats of the driver's bench command
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/bench_n1 -- python bench.py --steps 20 --warmup 5 --no_cpu_baseline > $OUT/bench_n1.json 2> $OUT/bench_n1.err
# 2. the same bench line without the profiler (the number to quote)
python bench.py --steps 20 --warmup 5 > $OUT/bench_line.json 2> $OUT/bench_line.err
# 3. per-rank shape of an 8-way expert-parallel run (8 local experts x 1024 rows): kernel stats
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/bench_ep8shape -- python bench.py --steps 20 --warmup 5 --experts 8 --no_cpu_baseline --no_extra > $OUT/bench_ep8shape.json 2> $OUT/bench_ep8shape.err
# 4. PMC passes on the ping-pong GEMM at the two EP-8 per-rank shapes
GEMM_TILE=4 python tools/pmc_run.py $OUT/pmc_pp_2048 expert_gemm -- python tools/gemm_pmc_probe.py 8 1024 2048 2048 > $OUT/pmc_pp_2048.txt 2>&1
GEMM_TILE=4 python tools/pmc_run.py $OUT/pmc_pp_4096 expert_gemm -- python tools/gemm_pmc_probe.py 8 1024 4096 4096 > $OUT/pmc_pp_4096.txt 2>&1
# 5. HBM traffic of the headline fc1 launch (separate FETCH / WRITE passes)
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python bench.py --steps 20 --warmup 5 --settle 20 --no_cpu_baseline --no_extra > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python bench.py --steps 20 --warmup 5 --settle 20 --no_cpu_baseline --no_extra > /dev/null 2>&1
python - <<'PY'
import csv, glob, collections
for name in ("fetch", "write"):
    acc = collections.defaultdict(lambda: [0.0, 0])
    for f in glob.glob(f"gpurun_out/prof_r02/{name}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = (r["Kernel_Name"][:80], r["Counter_Name"])
            acc[k][0] += float(r["Counter_Value"]); acc[k][1] += 1
    with open(f"gpurun_out/prof_r02/{name}_summary.txt", "w") as fo:
        for (k, c), (s, n) in sorted(acc.items(), key=lambda kv: -kv[1][0])[:12]:
            fo.write(f"{k:80s} {c:12s} avg {s / n:14.1f} (n={n})\n")
PY
find $OUT -name "*kernel_stats.csv" | head
