#!/bin/bash
# Round-6 profile collection on the GPU box (run through gpurun from the repo root):  bash tools/profile_r06.sh [quick]
# Writes rocprofv3 summaries under gpurun_out/prof_r06/ and gpurun_out/prof_r06/traffic.json (stamped with the sha256 of the
# kernel source it was measured on); the files worth keeping are copied to profiles/ by hand.
set -x
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_r06
mkdir -p $OUT
# 1. kernel trace + stats of the driver's bench command (eager: one kernel record per launch either way)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/bench_n1 -- python bench.py --steps 20 --warmup 5 --no_cpu_baseline --no_extra > $OUT/bench_n1.json 2> $OUT/bench_n1.err
# 2. the same bench line without the profiler (the number to quote)
python bench.py --steps 20 --warmup 5 > $OUT/bench_line.json 2> $OUT/bench_line.err
# 3. HBM traffic of the headline fc1 launch (separate FETCH / WRITE passes, as the guide prescribes)
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python bench.py --eager --steps 20 --warmup 5 --settle 20 --no_cpu_baseline --no_extra > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python bench.py --eager --steps 20 --warmup 5 --settle 20 --no_cpu_baseline --no_extra > /dev/null 2>&1
python - <<'PY'
import csv, glob, collections, hashlib, json, os
out = "gpurun_out/prof_r06"
FC1 = "expert_gemm_big_kernel<bf16_t, true, 1, 4, 3, true, 128, true>"   # the 128 x 256 ring kernel, fused ReLU epilogue + fused location = fc1
val = {}
for name in ("fetch", "write"):
    acc = collections.defaultdict(lambda: [0.0, 0])
    for f in glob.glob(f"{out}/{name}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = (r["Kernel_Name"][:100], r["Counter_Name"])
            acc[k][0] += float(r["Counter_Value"]); acc[k][1] += 1
    with open(f"{out}/{name}_summary.txt", "w") as fo:
        for (k, c), (s, n) in sorted(acc.items(), key=lambda kv: -kv[1][0])[:12]:
            fo.write(f"{k:100s} {c:12s} avg {s / n:14.1f} (n={n})\n")
            if FC1 in k:
                val[name] = (k, s / n, n)
avg_us, kern, dec_us, dec_kern = None, None, None, None
for st in glob.glob(f"{out}/bench_n1/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(st)):
        if FC1 in r["Name"]:
            avg_us, kern = float(r["AverageNs"]) * 1e-3, r["Name"]
        if "decode_kernel<bf16_t, 2," in r["Name"] or "decode_fin_kernel<bf16_t, 2>" in r["Name"]:
            dec_us, dec_kern = float(r["AverageNs"]) * 1e-3, r["Name"][:60]
h = hashlib.sha256()
for f in ("expert_gemm.hip", "gemm_dev.h", "common.h"):
    h.update(open(os.path.join("tutel_amd", "csrc", f), "rb").read())
fv, wv = val.get("fetch", (None, None, 0))[1], val.get("write", (None, None, 0))[1]
tj = {"expert_gemm_hip_sha256": h.hexdigest(), "git_head": os.environ.get("GIT_HEAD", "unknown"), "kernel": kern,
      "expert_gemm_fc1_hbm_bytes_per_launch": int(fv * 2 * 1024 + wv * 1024) if fv and wv else None,
      "fetch_size_kb_avg": fv, "write_size_kb_avg": wv, "expert_gemm_fc1_avg_us_rocprofv3": avg_us,
      "fast_decode_avg_us_rocprofv3": dec_us, "fast_decode_kernel": dec_kern,
      "source": "tools/profile_r06.sh: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes (gfx950: FETCH_SIZE x 2, KB -> bytes) and "
                "rocprofv3 --kernel-trace --stats of `python bench.py --steps 20 --warmup 5`"}
json.dump(tj, open(f"{out}/traffic.json", "w"), indent=1)
print(json.dumps(tj, indent=1))
PY
find $OUT -name "*kernel_stats.csv" | head
[ "$1" = "quick" ] && exit 0   # (re-stamping traffic.json after a source change that leaves the dominant kernel's loop alone)
# 4. per-rank shapes of the expert-parallel configurations, emulated on one GPU (bench lines only; kernel stats for the 8-expert one)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/bench_ep8shape -- python bench.py --steps 20 --warmup 5 --experts 8 --no_cpu_baseline --no_extra > $OUT/bench_ep8shape.json 2> $OUT/bench_ep8shape.err
