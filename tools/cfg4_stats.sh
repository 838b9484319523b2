#!/bin/bash
# rocprofv3 kernel stats of the configs[4] bf16 bench run (B=32).  usage: tools/cfg4_stats.sh TAG [ENV...]
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/cfg4; mkdir -p $O
rm -rf /tmp/prof_$TAG
env "$@" timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/prof_$TAG -o cfg4 --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --hidden 768 --word-heads 8 --window 5 --gsl-rate 0.8 --no-cpu-baseline --no-series --steps 10 --warmup 3 --batch 32 --gemm-mode bf16 --no-profile > $O/rocprof_$TAG.log 2>&1
f=$(find /tmp/prof_$TAG -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cp $f $O/cfg4_bf16_kernel_stats_$TAG.csv
python - <<P
import csv
rows = list(csv.DictReader(open("$O/cfg4_bf16_kernel_stats_$TAG.csv")))
for r in rows[:12]:
    print("  %-64s calls %5s avg_us %9.1f pct %5s" % (r["Name"][:64], r["Calls"], float(r["AverageNs"]) / 1e3, r["Percentage"]))
P
tail -1 $O/rocprof_$TAG.log | cut -c1-200
