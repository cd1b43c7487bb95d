#!/bin/bash
# round 5: tower GEMM with the activation operand pre-split (gemm_bxp_kernel<true>) against the in-loop split
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05r
mkdir -p $O
timeout 600 python profiles/ubench/gemm_presplit_ab.py > $O/gemm_presplit_ab.txt 2>&1; echo "lab exit $?"
cat $O/gemm_presplit_ab.txt
