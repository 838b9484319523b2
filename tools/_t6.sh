cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r6
C4="--hidden 768 --word-heads 8 --window 5 --gsl-rate 0.8 --gemm-mode bf16 --no-strong --no-other-configs --no-series --no-side-modes --no-cpu-baseline"
for rep in 1 2; do
for v in measure ppnt; do
python bench.py --measure-lib $v $C4 2>/dev/null | tail -1 | python -c "
import sys, json
d=json.loads(sys.stdin.read()); k=d['kernels']; print('$v', round(d['value']), round(d['roofline']['frac'],4), ' '.join('%s %.3f' % (n, k[n]['ms_per_step']) for n in ('gemm_big','gemm_big_tn','spmm')))"
done
done
