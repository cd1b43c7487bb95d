#!/bin/bash
# the forward with bf16 planes split once per tile (RBX_ATTN_STREAM=2) against the streamed f32 tiles (1), alone
mkdir -p gpurun_out/r04t
cd /root/repo
for S in 1 2 2; do RBX_ATTN_STREAM=$S timeout 120 profiles/ubench/attn_stream 200 4096 0 2>&1 | grep -v launch; done > gpurun_out/r04t/planes.txt 2>&1
RBX_ATTN_STREAM=2 timeout 120 profiles/ubench/attn_stream 200 4095 0 2>&1 | grep -v launch >> gpurun_out/r04t/planes.txt
RBX_ATTN_STREAM=2 timeout 120 profiles/ubench/attn_stream 224 1000 0 2>&1 | grep -v launch >> gpurun_out/r04t/planes.txt
cat gpurun_out/r04t/planes.txt
