#!/bin/bash
# pairs/s at the realistic evidence-count distribution (host-bound regime) under different environment settings
cd $GRAFT_REPO_ROOT
for e in "$@"; do
  if [ "$e" = "-" ]; then e="GET_AMD_AB_NONE=1"; fi
  for rep in 1 2; do
    env $e python bench.py --measure-build --steps 40 --warmup 10 --no-cpu-baseline --no-series --no-profile --evd-dist snopes 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$e', 'snopes pairs/s %.0f ms/step %.4f claims/s %.0f' % (d['value'], d['ms_per_step'], d['claims_per_s']))"
  done
done
