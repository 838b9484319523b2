#!/bin/bash
# BASELINE configs[4] (h=768, 8 word heads, window 5, gsl_rate 0.8): fp32 vs the bf16 storage pipeline, bench lines + kernel stats.
TAG=${1:-x}
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/cfg4; mkdir -p $O
C4="--hidden 768 --word-heads 8 --window 5 --gsl-rate 0.8 --no-cpu-baseline --no-series --steps 10 --warmup 3"
for B in 8 32; do
  for M in fp32 bf16; do
    timeout 300 python bench.py $C4 --batch $B --gemm-mode $M > $O/bench_${M}_b${B}_$TAG.json 2> $O/bench_${M}_b${B}_$TAG.err || tail -5 $O/bench_${M}_b${B}_$TAG.err
  done
done
cd /tmp
for M in fp32 bf16; do
  rm -rf /tmp/prof_$M
  timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/prof_$M -o cfg4 --output-format csv -- python $GRAFT_REPO_ROOT/bench.py $C4 --batch 32 --gemm-mode $M --no-profile > $O/rocprof_$M.log 2>&1
  f=$(find /tmp/prof_$M -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp $f $O/cfg4_${M}_kernel_stats_$TAG.csv
done
cd $GRAFT_REPO_ROOT
python - <<P
import json, csv
for B in (8, 32):
    for M in ("fp32", "bf16"):
        try:
            d = json.load(open("$O/bench_%s_b%d_$TAG.json" % (M, B)))
            print(M, "B=%d" % B, "pairs/s %.0f ms/step %.3f" % (d["value"], d["ms_per_step"]), json.dumps(d.get("roofline"))[:300])
        except Exception as e:
            print(M, B, "FAILED", e)
for M in ("fp32", "bf16"):
    try:
        rows = list(csv.DictReader(open("$O/cfg4_%s_kernel_stats_$TAG.csv" % M)))
        print("--", M)
        for r in rows[:14]:
            print("  %-70s calls %6s avg_us %9.1f pct %5s" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e3, r["Percentage"]))
    except Exception as e:
        print(M, "stats FAILED", e)
P
