#!/bin/bash
# per-kernel table (ms/step, achieved rate) of the default bench workload under different environment settings.
# usage: tools/ab_kern.sh "ENV1" "ENV2" ...  ("-" = none)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2
for e in "$@"; do
  if [ "$e" = "-" ]; then e="GET_AMD_AB_NONE=1"; fi
  env $e python bench.py --measure-build --steps 20 --warmup 6 --no-cpu-baseline --no-series 2> gpurun_out/r2/ab_err.log | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('== $e', 'pairs/s %.0f  ms/step %.4f  parity %s mism %s' % (d['value'], d['ms_per_step'], d.get('parity', {}).get('max_abs_logit_diff_vs_cpu_oracle'), d.get('parity', {}).get('graphs_with_real_node_keep_set_mismatch')))
for k, v in d.get('kernels', {}).items():
    r = v.get('achieved_gbps', v.get('achieved_tflops', 0))
    print('   %-16s %7.4f ms/step  %5.1f launches  %8.1f %s  frac %.3f' % (k, v['ms_per_step'], v['launches_per_step'], r, 'GB/s' if 'achieved_gbps' in v else 'TF', v['frac']))
" || tail -5 gpurun_out/r2/ab_err.log
done
