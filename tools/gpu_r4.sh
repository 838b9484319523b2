#!/bin/bash
# Round-4 GPU pass on one box: GPU tests (optionally -k EXPR), the default bench line, summary.  usage: tools/gpu_r4.sh TAG [pytest -k expr] [bench args...]
TAG=${1:-x}; KEXPR=${2:-}; shift; shift
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4; mkdir -p $O
if [ "$KEXPR" != "none" ]; then
  if [ -n "$KEXPR" ]; then
    timeout 2400 python -m pytest tests -m gpu -q -x -k "$KEXPR" 2>&1 | tail -40 > $O/pytest_$TAG.log
  else
    timeout 2400 python -m pytest tests -m gpu -q -x 2>&1 | tail -40 > $O/pytest_$TAG.log
  fi
  tail -8 $O/pytest_$TAG.log
fi
timeout 1200 python bench.py "$@" > $O/bench_$TAG.json 2> $O/bench_$TAG.err; tail -3 $O/bench_$TAG.err
python tools/bench_summary.py $O/bench_$TAG.json
