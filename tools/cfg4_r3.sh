#!/bin/bash
# configs[4] (h = 768) bf16: tests + bench kernel table.  usage: tools/cfg4_r3.sh
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -q -x -k "bf16" 2>&1 | tail -3
python bench.py --hidden 768 --word-heads 8 --window 5 --gsl-rate 0.8 --batch 32 --no-cpu-baseline --no-series --gemm-mode bf16 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['avg_launch_ms'])
for k,v in d['kernels'].items(): print('  %-16s %7.4f %5.1f %8.1f %.3f'%(k,v['ms_per_step'],v['launches_per_step'],v.get('achieved_gbps',v.get('achieved_tflops',0)),v['frac']))
"
