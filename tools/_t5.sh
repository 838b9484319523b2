cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r6
C4="--hidden 768 --word-heads 8 --window 5 --gsl-rate 0.8 --gemm-mode bf16 --no-strong --no-other-configs --no-series --no-side-modes --no-cpu-baseline"
for e in GH_BF16_TILE=1 GH_X=1 GH_BF16_TILE=1 GH_X=1; do
env $e python bench.py --measure-build $C4 2>/dev/null | tail -1 | python -c "
import sys, json
d=json.loads(sys.stdin.read()); print('$e', d['value'], d['roofline']['frac'], d.get('parity'))"
done
