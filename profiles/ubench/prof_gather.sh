#!/bin/bash
# rocprofv3 kernel stats of the two sequence-gather shapes, for each library variant named on the command
# line ("base" = the shipped library).  Run on the GPU box from the repo root.
export TMPDIR=/tmp
for v in "${@:-base}"; do
  out=/root/repo/gpurun_out/profg_$v
  rm -rf $out
  if [ "$v" != base ]; then export RECBOX_HIP_LIB=/root/repo/recbox_amd/lib/librecbox_hip_$v.so; else unset RECBOX_HIP_LIB; fi
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $out -o kb -- python /root/repo/profiles/ubench/kernels_bench.py gather > $out.log 2>&1)
  echo "== $v"
  python /root/repo/profiles/topk.py $(find $out -name "*.db" | head -1) | grep -E "embed_seq|segment_reduce"
done
