#!/bin/bash
# round 5: gemm_bxs_kernel (halves of the workgroup a phase apart) against gemm_bxp_kernel
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05u
mkdir -p $O
timeout 300 python profiles/ubench/gemm_stagger_ab.py > $O/gemm_stagger_ab.txt 2>&1; echo "lab exit $?"
cat $O/gemm_stagger_ab.txt
timeout 600 python -m pytest tests/test_gpu_matching.py -q -m gpu -x -k "gemm or tower or linear" 2>&1 | tail -4
