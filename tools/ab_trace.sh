#!/bin/bash
# A/B the per-dispatch step timeline of several library builds on ONE box.  usage: tools/ab_trace.sh TAG lib1.so lib2.so ...
TAG=$1; shift
cd $GRAFT_REPO_ROOT
i=0
for lib in "$@"; do
  for rep in a b; do
    GET_AMD_LIB=$GRAFT_REPO_ROOT/$lib bash tools/trace_step.sh ${TAG}_${i}${rep} > /dev/null 2>&1
  done
  i=$((i+1))
done
python - "$TAG" "$#" <<'P'
import sys, re
tag, n = sys.argv[1], int(sys.argv[2])
cols = []
for i in range(n):
    for rep in "ab":
        rows = [l for l in open(f"gpurun_out/r2/trace_{tag}_{i}{rep}.txt") if " us " in l]
        cols.append(rows)
m = min(len(c) for c in cols)
def parse(l):
    f = l.split()
    return float(f[5]), int(f[8]), " ".join(f[9:])[:44]
print("  ".join(f"lib{i}{r}" for i in range(n) for r in "ab"))
for k in range(m):
    vals = [parse(c[k]) for c in cols]
    if max(v[0] for v in vals) < 25: continue
    print("  ".join("%7.1f" % v[0] for v in vals), " grid %5d  %s" % (vals[0][1], vals[0][2]))
for i, c in enumerate(cols):
    print("lib%d%s %s" % (i // 2, "ab"[i % 2], [l for l in open(f"gpurun_out/r2/trace_{tag}_{i//2}{'ab'[i%2]}.txt")][-1].strip()))
P
