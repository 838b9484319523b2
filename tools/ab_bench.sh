#!/bin/bash
# pairs/s of the default bench workload under different environment settings.  usage: tools/ab_bench.sh "ENV1" "ENV2" ...  ("-" = none)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6
for e in "$@"; do
  if [ "$e" = "-" ]; then e="GET_AMD_AB_NONE=1"; fi
  for rep in 1 2; do
    env $e python bench.py --measure-build --steps 30 --warmup 8 --no-cpu-baseline --no-series --no-profile $AB_ARGS 2> gpurun_out/r6/ab_err.log | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$e', 'pairs/s %.0f  ms/step %.4f  parity %s' % (d['value'], d['ms_per_step'], d.get('parity', {}).get('max_abs_logit_diff_vs_cpu_oracle')))
" || tail -5 gpurun_out/r6/ab_err.log
  done
done
