#!/bin/bash
# the streamed attention forward kernel against the resident one, alone (profiles/ubench/attn_stream.hip)
mkdir -p gpurun_out/r04n
cd /root/repo
for L in 200 96 224; do
  for S in 0 1; do
    RBX_ATTN_STREAM=$S timeout 120 profiles/ubench/attn_stream $L 4096 0 2>&1 | tail -3
  done
done > gpurun_out/r04n/attn_stream.txt 2>&1
RBX_ATTN_STREAM=1 timeout 120 profiles/ubench/attn_stream 200 4095 0 >> gpurun_out/r04n/attn_stream.txt 2>&1
cat gpurun_out/r04n/attn_stream.txt
