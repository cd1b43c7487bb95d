#!/bin/bash
# the attention kernels with their tile products on the bf16 matrix cores (operands split three ways, six products) against
# the f32 MFMA form, alone: resident and streamed forward, both backward kernels; L = 200 and 50, d = 64
mkdir -p gpurun_out/r04q
cd /root/repo
for bin in attn_stream_f32 attn_stream; do
  for S in 0 1; do
    echo "== $bin"; RBX_ATTN_STREAM=$S RBX_ATTN_PREFETCH=${PF:-1} timeout 120 profiles/ubench/$bin 200 4096 1 2>&1 | grep -v launch
  done
done > gpurun_out/r04q/bf16x6.txt 2>&1
echo "== attn_stream, no prefetch loop" >> gpurun_out/r04q/bf16x6.txt
RBX_ATTN_STREAM=0 RBX_ATTN_PREFETCH=0 timeout 120 profiles/ubench/attn_stream 200 4096 1 2>&1 | grep -v launch >> gpurun_out/r04q/bf16x6.txt
for bin in attn_stream_f32 attn_stream; do echo "== $bin L=50"; timeout 120 profiles/ubench/$bin 50 8192 1 2>&1 | grep -v launch; done >> gpurun_out/r04q/bf16x6.txt
cat gpurun_out/r04q/bf16x6.txt
