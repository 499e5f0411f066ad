#!/bin/bash
# round 6: HBM traffic counters of the persistent fc1 -> fc2 launch (TUTEL_AMD_FFN_FUSED=1), separate FETCH_SIZE / WRITE_SIZE passes
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r06_ffn_pmc; mkdir -p $OUT
for c in FETCH_SIZE WRITE_SIZE; do
  TUTEL_AMD_FFN_FUSED=1 rocprofv3 --pmc $c --output-format csv -d $OUT/$c -- python bench.py --eager --steps 20 --warmup 5 --settle 20 --no_cpu_baseline --no_extra > /dev/null 2>&1
done
python3 - <<'PY'
import csv, glob, collections
out = "gpurun_out/r06_ffn_pmc"
res = {}
for name in ("FETCH_SIZE", "WRITE_SIZE"):
    acc = collections.defaultdict(lambda: [0.0, 0])
    for f in glob.glob(f"{out}/{name}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            acc[r["Kernel_Name"][:60]][0] += float(r["Counter_Value"]); acc[r["Kernel_Name"][:60]][1] += 1
    for k, (s, n) in acc.items():
        if "expert_ffn_kernel" in k:
            res[name] = (s / n, n)
fv, wv = res["FETCH_SIZE"][0], res["WRITE_SIZE"][0]
alg = (2 * 64 * 2048 * 2048 + 64 * 128 * 2048 * 2) * 2          # weights of both GEMMs + token rows in + output rows out (hidden on chip)
two = alg + 2 * 64 * 128 * 2048 * 2                              # + the hidden activation written and read back
tot = fv * 2 * 1024 + wv * 1024
with open(f"{out}/summary.txt", "w") as f:
    f.write(f"expert_ffn_kernel<bf16_t, true> (TUTEL_AMD_FFN_FUSED=1, headline shape), averages over {res['FETCH_SIZE'][1]} / {res['WRITE_SIZE'][1]} launches\n")
    f.write(f"FETCH_SIZE avg {fv:.1f} KB (x 2 on gfx950), WRITE_SIZE avg {wv:.1f} KB -> {tot / 1e6:.1f} MB per launch\n")
    f.write(f"algorithmic, hidden activation on chip (SURVEY 8d): {alg / 1e6:.1f} MB; with the hidden activation written and read once: {two / 1e6:.1f} MB\n")
    f.write(f"traffic / (algorithmic + hidden round trip) = {tot / two:.4f}\n")
print(open(f"{out}/summary.txt").read())
PY
