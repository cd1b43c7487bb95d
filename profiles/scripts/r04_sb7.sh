#!/bin/bash
out=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $out
python -m pytest tests/test_gpu_matching.py tests/test_gpu_ranking.py -q -k "deepfm or DeepFM" 2>&1 | tail -4 > $out/sb_tests.log
B="--config deepfm --steps 20 --warmup 5 --no-cpu-baseline"
for i in 1 2; do
timeout 300 python bench.py $B > $out/dfm_fused_$i.json 2> /dev/null
RECBOX_AMD_FUSE_DEEPFM_LR=0 timeout 300 python bench.py $B > $out/dfm_off_$i.json 2> /dev/null
done
for f in dfm_fused_1 dfm_off_1 dfm_fused_2 dfm_off_2; do echo $f $(python -c "import json,sys; d=json.load(open('$out/$f.json')); print(d['ms_per_step'])"); done
grep -E "passed|failed" $out/sb_tests.log
