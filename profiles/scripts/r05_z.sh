#!/bin/bash
# round 5: embed_seq_kernel's parameters at cfg 3's shape (rows in flight per lane, lanes per sample, occupancy bound)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05z
mkdir -p $O
for v in default seq_u2 seq_u6 seq_u8 seq_sg32 seq_sg32u8 seq_w8 default; do
  if [ $v != default ]; then export RECBOX_HIP_LIB=$GRAFT_REPO_ROOT/recbox_amd/lib/librecbox_hip_$v.so; else unset RECBOX_HIP_LIB; fi
  timeout 200 python profiles/ubench/seq_gather_lab.py 2>&1 | grep "us " | tee -a $O/seq_gather_lab.txt
done
