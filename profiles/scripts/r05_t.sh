#!/bin/bash
# round 5: gemm_bxw_kernel (column strips) against gemm_bxp_kernel
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05t
mkdir -p $O
timeout 600 python profiles/ubench/gemm_strips_ab.py > $O/gemm_strips_ab.txt 2>&1; echo "lab exit $?"
cat $O/gemm_strips_ab.txt
timeout 600 python -m pytest tests/test_gpu_matching.py -q -m gpu -x -k "gemm or tower or linear" 2>&1 | tail -4
