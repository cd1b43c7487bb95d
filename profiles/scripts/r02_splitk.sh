#!/bin/bash
out=/root/repo/gpurun_out/r2sk
rm -rf $out; mkdir -p $out
export TMPDIR=/tmp
cd /root/repo
for v in 2 4; do
  export RBX_GEMM_SPLIT_WGS=$v
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $out/prof -o b -- python /root/repo/bench.py --config deepfm --no-cpu-baseline --steps 10 --warmup 3 > $out/prof_$v.log 2>&1)
  python profiles/topk.py $(find $out/prof -name "*.db" | head -1) 40 > $out/stats_$v.txt
  rm -rf $out/prof
  echo SPLIT_WGS_PER_CU=$v $(grep -o '"ms_per_step": [0-9.]*' $out/prof_$v.log); grep "gemm_f32_kernel<false, false>\|splitk" $out/stats_$v.txt | cut -c1-140
done
unset RBX_GEMM_SPLIT_WGS
timeout 900 python -m pytest tests/test_gpu_matching.py -x -q -m gpu -k "linear or mlp or tower or deepfm or dssm or youtube" 2>&1 | tail -2
for c in youtubednn sasrec; do timeout 600 python bench.py --config $c --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('$c step_ms', round(d['ms_per_step'],3))"; done
