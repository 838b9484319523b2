#!/bin/bash
# slabs-per-workgroup sweep of the aggregation kernel inside the bench step.  usage: tools/ab_spw.sh "bench flags" spw1 spw2 ...
cd $GRAFT_REPO_ROOT
FL="$1"; shift
for spw in "$@"; do
  GH_SPMM_SPW=$spw python bench.py --measure-build $FL --steps 10 --warmup 4 --no-cpu-baseline --no-series --no-side-modes 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
v = d['kernels']['spmm']
print('spw $spw   pairs/s %.0f  ms/step %.4f   spmm %.4f ms/step %.1f launches %.1f GB/s' % (d['value'], d['ms_per_step'], v['ms_per_step'], v['launches_per_step'], v['achieved_gbps']))"
done
