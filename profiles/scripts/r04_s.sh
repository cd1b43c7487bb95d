#!/bin/bash
# the attention forms of this round in the test suite and in the SASRec step
out=/root/repo/gpurun_out/r04s
mkdir -p $out
cd /root/repo
S=1 BINS=attn_stream bash profiles/scripts/r04_r.sh > $out/kernels_alone.txt 2>&1
timeout 900 python -m pytest tests/test_gpu_matching.py tests/test_gpu_cabi_vs_c_oracle.py -m gpu -x -q -k "attn or attention or sasrec or SASRec" > $out/tests.log 2>&1
tail -5 $out/tests.log
for S in 0 1; do
  RBX_ATTN_STREAM=$S timeout 300 python bench.py --config sasrec --no-cpu-baseline --no-extra-configs --steps 30 --warmup 5 2>/dev/null | tail -1 > $out/bench_sasrec_stream$S.json
  python -c "import json;d=json.load(open('$out/bench_sasrec_stream$S.json'));print('stream',$S,d['ms_per_step'])"
done
cat $out/kernels_alone.txt
