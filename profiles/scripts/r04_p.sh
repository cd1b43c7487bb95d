#!/bin/bash
# s_setprio around the tile products of the attention kernels (forward resident / streamed, both backward kernels), alone
mkdir -p gpurun_out/r04p
cd /root/repo
for rep in 1 2; do
for bin in attn_stream attn_stream_prio; do
  for S in 0 1; do
    echo "== $bin"; RBX_ATTN_STREAM=$S timeout 120 profiles/ubench/$bin 200 4096 1 2>&1 | grep -v launch
  done
done
done > gpurun_out/r04p/prio.txt 2>&1
cat gpurun_out/r04p/prio.txt
