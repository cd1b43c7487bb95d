#!/bin/bash
# round 4: tier C second form (partition pass + bucket reduce): tests, bench on / off, timeline; DeepFM training test detail
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/r04h
mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_ranking.py tests/test_gpu_optim.py tests/test_gpu_cabi_vs_c_oracle.py -q -m gpu -x > $out/tests_fm.log 2>&1
echo "fm tests exit $?" | tee -a $out/summary.txt; tail -4 $out/tests_fm.log | tee -a $out/summary.txt
timeout 600 python -m pytest tests/test_gpu_matching.py -q -m gpu -k "training_step or benchmarked_batch" > $out/tests_parity.log 2>&1
echo "parity tests exit $?" | tee -a $out/summary.txt; grep -E "passed|failed|grad|max abs" $out/tests_parity.log | head -40 | tee -a $out/summary.txt
RECBOX_AMD_GEMM_BX6=0 RBX_GEMM_BX6=0 timeout 600 python -m pytest tests/test_gpu_matching.py -q -m gpu -k "training_step" > $out/tests_parity_f32.log 2>&1
echo "parity tests (f32 MFMA GEMMs) exit $?" | tee -a $out/summary.txt; grep -E "passed|failed|grad|max abs" $out/tests_parity_f32.log | head -40 | tee -a $out/summary.txt
ms() { python -c "import json,sys; d=json.loads([l for l in open('$out/bench_$1.json') if l.startswith('{')][-1]); print('$1', round(d['ms_per_step'],4), (d.get('roofline') or {}).get('kernel_ms'))" 2>&1 | tail -1 | tee -a $out/summary.txt; }
timeout 300 python bench.py --steps 50 --warmup 10 --no-extra-configs --no-cpu-baseline > $out/bench_fm.json 2> $out/bench_fm.err; ms fm
RBX_FM_TIER_C=0 timeout 300 python bench.py --steps 50 --warmup 10 --no-extra-configs --no-cpu-baseline > $out/bench_fm_tierc_off.json 2>/dev/null; ms fm_tierc_off
timeout 300 python bench.py --steps 50 --warmup 10 --no-extra-configs --no-cpu-baseline --dist zipf > $out/bench_fm_zipf.json 2>/dev/null; ms fm_zipf
RBX_FM_TIER_C=0 timeout 300 python bench.py --steps 50 --warmup 10 --no-extra-configs --no-cpu-baseline --dist zipf > $out/bench_fm_zipf_tierc_off.json 2>/dev/null; ms fm_zipf_tierc_off
RECBOX_AMD_FM_BLOCKSORT_AT=side RECBOX_AMD_FM_REZERO_ON=main timeout 300 python bench.py --steps 50 --warmup 10 --no-extra-configs --no-cpu-baseline > $out/bench_fm_blocksort_side_rezero_main.json 2>/dev/null; ms fm_blocksort_side_rezero_main
RECBOX_AMD_FM_BLOCKSORT_AT=side timeout 300 python bench.py --steps 50 --warmup 10 --no-extra-configs --no-cpu-baseline > $out/bench_fm_blocksort_side.json 2>/dev/null; ms fm_blocksort_side
prof() { # name, bench args, anchor kernel, occurrence
  rm -rf $out/prof
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $out/prof -o b -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-extra-configs $2 > $out/prof_$1.log 2>&1)
  db=$(find $out/prof -name "*.db" | head -1)
  python profiles/topk.py $db 25 > $out/$1_kernel_stats.txt
  python profiles/timeline.py $db "$3" $4 > $out/$1_replay_timeline.txt 2>&1
  rm -rf $out/prof
}
prof fm "--steps 20 --warmup 5" compact_ids 30
BS=side; (export RECBOX_AMD_FM_BLOCKSORT_AT=side; prof fm_blocksort_side "--steps 20 --warmup 5" compact_ids 30)
