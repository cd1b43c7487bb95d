#!/bin/bash
# the planes forward with parts compiled out (RBX_PL_ABL: 1 no tile streaming in the loop, 2 no tile steps, 3 both = barriers + stores)
mkdir -p gpurun_out/r04u
cd /root/repo
for a in 0 1 2 3; do echo "ABL=$a"; RBX_ATTN_STREAM=2 timeout 120 profiles/ubench/attn_stream_x$a 200 4096 0 2>&1 | grep forward; done > gpurun_out/r04u/planes_parts.txt 2>&1
cat gpurun_out/r04u/planes_parts.txt
