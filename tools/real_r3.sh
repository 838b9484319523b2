#!/bin/bash
# realistic-regime check: full GPU tests + the snopes bench at B = 32 / 64 and the headline.  usage: tools/real_r3.sh
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -3
for a in "--evd-dist snopes" "--batch 64 --evd-dist snopes" ""; do
python bench.py $a --no-cpu-baseline --no-series --no-side-modes --no-profile 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$a', round(d['value']), round(d['ms_per_step'],3), d['timed']['blocks'], d['timed']['spread_rel'])"
done
