cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r6
for rep in 1 2; do
VARIANT=e32 tools/ab_e32.sh GH_DBG=128 GH_X=pd2 | awk 'NR<=2'
VARIANT=e32pd1 tools/ab_e32.sh GH_X=pd1 | awk 'NR<=1'
VARIANT=e32nosb tools/ab_e32.sh GH_X=nosb | awk 'NR<=1'
done
