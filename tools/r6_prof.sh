#!/bin/bash
# rocprofv3 kernel stats of the bench command (round 6).  usage: tools/r6_prof.sh <tag> [env assignments...]
# writes gpurun_out/prof_<tag>/..._kernel_stats.csv and prints the top kernels
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/prof_$tag
mkdir -p $out
env "$@" rocprofv3 --kernel-trace --stats --output-format csv -d $out -o r -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no_extra --no_cpu_baseline > $out/bench.json 2> $out/bench.err
f=$(find $out -name "*kernel_stats.csv" | head -1)
cp "$f" $GRAFT_REPO_ROOT/gpurun_out/prof_${tag}_kernel_stats.csv
python3 - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows[:9]:
    print("%-90s calls %5s avg %9.2f us  tot%% %s" % (r["Name"][:90], r["Calls"], float(r["AverageNs"])/1e3, r["Percentage"]))
PY
rm -rf $out/*/  # traces are large; the stats csv was copied
