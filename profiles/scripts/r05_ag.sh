#!/bin/bash
# round 5: gemm_bxp_kernel with its accumulators in AccVGPRs (inline-asm MFMAs, "+a") against the compiler's VGPR form
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05ag
mkdir -p $O
for rep in 1 2; do
for v in default agpr; do
  if [ $v != default ]; then export RECBOX_HIP_LIB=$GRAFT_REPO_ROOT/recbox_amd/lib/librecbox_hip_$v.so; else unset RECBOX_HIP_LIB; fi
  timeout 300 python profiles/ubench/gemm_shapes.py 2>&1 | grep -v "amdgpu\|UserWarning\|Consider using\|% (M, K" | tee -a $O/gemm_shapes.txt
done
done
