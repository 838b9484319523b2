#!/bin/bash
# per-dispatch step timelines under different environment settings, side by side.  usage: tools/ab_env.sh TAG "ENV1" "ENV2" ...   (ENV like GH_DBG=1; "-" = none)
TAG=$1; shift
cd $GRAFT_REPO_ROOT
i=0
for e in "$@"; do
  if [ "$e" = "-" ]; then bash tools/trace_step.sh ${TAG}_$i > /dev/null 2>&1; else env $e bash tools/trace_step.sh ${TAG}_$i > /dev/null 2>&1; fi
  i=$((i+1))
done
python - "$TAG" "$#" <<'P'
import sys
tag, n = sys.argv[1], int(sys.argv[2])
cols = [[l for l in open(f"gpurun_out/r2/trace_{tag}_{i}.txt") if " us " in l] for i in range(n)]
m = min(len(c) for c in cols)
def parse(l):
    f = l.split()
    return float(f[5]), int(f[8]), " ".join(f[9:])[:44]
for k in range(m):
    vals = [parse(c[k]) for c in cols]
    if max(v[0] for v in vals) < 25: continue
    print("  ".join("%7.1f" % v[0] for v in vals), " grid %5d  %s" % (vals[0][1], vals[0][2]))
for i in range(n):
    print(i, [l for l in open(f"gpurun_out/r2/trace_{tag}_{i}.txt")][-1].strip())
P
