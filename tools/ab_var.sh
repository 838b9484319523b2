#!/bin/bash
# pairs/s of the default bench workload on several VARIANT builds (make -C get_amd/csrc variant NAME=x EXTRA=...), alternating on ONE box.
# usage: tools/ab_var.sh [-r REPS] [-e "ENV=.."] name1 name2 ...     extra bench args through AB_ARGS
REPS=2; ENVS="GET_AMD_AB_NONE=1"
while getopts "r:e:" o; do case $o in r) REPS=$OPTARG;; e) ENVS=$OPTARG;; esac; done; shift $((OPTIND-1))
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r6
for rep in $(seq 1 $REPS); do
  for v in "$@"; do
    env $ENVS python bench.py --measure-lib $v --steps 30 --warmup 8 --no-cpu-baseline --no-series --no-side-modes --no-strong $AB_ARGS 2> gpurun_out/r6/ab_err.log | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
k = d.get('kernels', {})
print('%-10s pairs/s %.0f  ms/step %.4f  parity %s  ' % ('$v', d['value'], d['ms_per_step'], d.get('parity', {}).get('max_abs_logit_diff_vs_cpu_oracle')) + '  '.join('%s %.3f' % (n, k[n]['ms_per_step']) for n in ('gemm_big', 'gemm_big_tn', 'spmm') if n in k))
" || tail -5 gpurun_out/r6/ab_err.log
  done
done
