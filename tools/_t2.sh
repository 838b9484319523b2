#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r6
echo "== tests on the e32 variant"
GET_AMD_LIB=$GRAFT_REPO_ROOT/get_amd/lib/libget_hip_e32.so timeout 1500 python -m pytest -x -q -m gpu tests/test_gpu_ops.py tests/test_gpu_model.py tests/test_gpu_fullsize_grads.py tests/test_gpu_wide_composite.py 2>&1 | tail -4
VARIANT=e32 tools/ab_e32.sh GH_DBG=128 GH_X=1
VARIANT=e32pd1 tools/ab_e32.sh GH_X=1
