#!/bin/bash
# round 5: gemm_bxs_kernel with parts compiled out (RBX_BXS_ABL: 1 no MFMAs, 2 no memory half, 4 no split / stores, 8 no fragment
# reads, 16 no loads, 12 = 4 + 8) -- timings only, the results are wrong by construction
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05v
mkdir -p $O
for v in base abl1 abl2 abl4 abl8 abl16 abl12; do
  if [ $v != base ]; then export RECBOX_HIP_LIB=$GRAFT_REPO_ROOT/recbox_amd/lib/librecbox_hip_$v.so; else unset RECBOX_HIP_LIB; fi
  echo "== $v" | tee -a $O/abl.txt
  timeout 200 python profiles/ubench/gemm_stagger_ab.py 2>&1 | grep "16384,4096\|65536,1677" | tee -a $O/abl.txt
done
