#!/bin/bash
# per-kernel durations of the attention kernels alone, f32 MFMA form against the bf16 six-product form
out=/root/repo/gpurun_out/r04r
mkdir -p $out
rm -f $out/kernels.txt
cd /tmp && export TMPDIR=/tmp
for bin in ${BINS:-attn_stream_f32 attn_stream}; do
  (RBX_ATTN_STREAM=${S:-0} timeout 120 rocprofv3 --kernel-trace --stats -d $out/kt_$bin -o b -- /root/repo/profiles/ubench/$bin 200 4096 1 > $out/kt_$bin.log 2>&1)
  db=$(find $out/kt_$bin -name "*.db" | head -1)
  echo "== $bin" >> $out/kernels.txt
  if [ -n "$db" ]; then timeout 60 python /root/repo/profiles/topk.py $db 6 < /dev/null 2>&1 | cut -c1-150 >> $out/kernels.txt; fi
  rm -rf $out/kt_$bin
done
cat $out/kernels.txt
