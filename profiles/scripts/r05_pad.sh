#!/bin/bash
# round 5: LDS plane stride of the split-operand GEMMs staggered by 64 / 16 bytes (the three planes of a B chunk on different banks)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05pad
mkdir -p $O
for rep in 1 2; do
for v in default pad32 pad8; do
  if [ $v != default ]; then export RECBOX_HIP_LIB=$GRAFT_REPO_ROOT/recbox_amd/lib/librecbox_hip_$v.so; else unset RECBOX_HIP_LIB; fi
  timeout 300 python profiles/ubench/gemm_shapes.py 2>&1 | grep -v amdgpu | tee -a $O/gemm_shapes.txt
done
done
