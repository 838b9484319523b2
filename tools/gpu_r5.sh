#!/bin/bash
# Round-5 GPU pass on one box.  usage: tools/gpu_r5.sh TAG PYTEST_K [bench args ...]   (PYTEST_K: "" = whole GPU suite, "none" = skip;
# bench args "none" = skip the bench; several bench runs: separate their argument lists with "::")
TAG=${1:-x}; KEXPR=${2:-}; shift; shift
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5; mkdir -p $O
if [ "$KEXPR" != "none" ]; then
  if [ -n "$KEXPR" ]; then
    timeout 2400 python -m pytest tests -m gpu -q -x -k "$KEXPR" 2>&1 | tail -60 > $O/pytest_$TAG.log
  else
    timeout 2400 python -m pytest tests -m gpu -q -x 2>&1 | tail -60 > $O/pytest_$TAG.log
  fi
  tail -25 $O/pytest_$TAG.log
fi
[ "$1" == "none" ] && exit 0
i=0; args=()
run_one() {
  timeout 1500 python bench.py "${args[@]}" > $O/bench_${TAG}_$i.json 2> $O/bench_${TAG}_$i.err; tail -3 $O/bench_${TAG}_$i.err
  python tools/bench_summary.py $O/bench_${TAG}_$i.json
  i=$((i+1)); args=()
}
for a in "$@"; do
  if [ "$a" == "::" ]; then run_one; else args+=("$a"); fi
done
run_one
