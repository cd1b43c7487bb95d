#!/bin/bash
# round 4 evidence, second edition (after the sequence-block chains of csrc/rbx_seqblock.hip), ONE call on one box: full GPU suite (parity ledger), the driver's bench line, world-of-one sharded lines,
# tier C on / off lines, rocprofv3 kernel stats + replay timelines, PMC traffic of the FM kernels (tier C on and off), smoke
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/r04
rm -rf $out; mkdir -p $out
rm -f gpurun_out/parity_errors.txt
timeout 1800 python -m pytest tests -q -m gpu > $out/gpu_tests.log 2>&1
echo "gpu tests exit $?" | tee -a $out/summary.txt; tail -3 $out/gpu_tests.log | tee -a $out/summary.txt
cp gpurun_out/parity_errors.txt $out/parity_errors.txt 2>/dev/null
ms() { python -c "import json,sys; d=json.loads([l for l in open('$out/bench_$1.json') if l.startswith('{')][-1]); r=d.get('roofline') or {}; print('$1', round(d['ms_per_step'],4), r.get('kernel_ms'), r.get('frac'))" 2>&1 | tail -1 | tee -a $out/summary.txt; }
timeout 900 python bench.py > $out/bench_fm.json 2> $out/bench_fm.err; ms fm
for cfg in youtubednn deepfm sasrec; do
  timeout 600 python bench.py --config $cfg --steps 20 --warmup 5 > $out/bench_$cfg.json 2>/dev/null; ms $cfg
done
for cfg in fm youtubednn deepfm; do
  timeout 300 python bench.py --config $cfg --force-sharded --steps 30 --warmup 5 --no-cpu-baseline > $out/bench_${cfg}_sharded1.json 2> $out/bench_${cfg}_sharded1.err; ms ${cfg}_sharded1
done
timeout 300 python bench.py --config fm --force-sharded --sharded-graph pieces --steps 30 --warmup 5 --no-cpu-baseline > $out/bench_fm_sharded1_pieces.json 2>/dev/null; ms fm_sharded1_pieces
RECBOX_AMD_DIRECT_RCCL=0 timeout 300 python bench.py --config fm --force-sharded --steps 30 --warmup 5 --no-cpu-baseline > $out/bench_fm_sharded1_torch_distributed.json 2>/dev/null; ms fm_sharded1_torch_distributed
timeout 300 python bench.py --config youtubednn --force-sharded --sharded-graph eager --steps 30 --warmup 5 --no-cpu-baseline > $out/bench_youtubednn_sharded1_eager.json 2>/dev/null; ms youtubednn_sharded1_eager
timeout 300 python bench.py --config deepfm --force-sharded --sharded-graph eager --steps 30 --warmup 5 --no-cpu-baseline > $out/bench_deepfm_sharded1_eager.json 2>/dev/null; ms deepfm_sharded1_eager
# SASRec with the sequence-block chains off (the two sub-layer nodes of the start of the round) and with only the forward chains
RECBOX_AMD_SEQBLOCK=0 timeout 600 python bench.py --config sasrec --steps 20 --warmup 5 --no-cpu-baseline > $out/bench_sasrec_seqblock_off.json 2>/dev/null; ms sasrec_seqblock_off
RECBOX_AMD_SEQBLOCK_BWD=0 timeout 600 python bench.py --config sasrec --steps 20 --warmup 5 --no-cpu-baseline > $out/bench_sasrec_seqblock_fwd_only.json 2>/dev/null; ms sasrec_seqblock_fwd_only
RBX_SB_BWD_ORDER=1 timeout 600 python bench.py --config sasrec --steps 20 --warmup 5 --no-cpu-baseline > $out/bench_sasrec_ffn_bwd_order1.json 2>/dev/null; ms sasrec_ffn_bwd_order1
# FM variants
RECBOX_AMD_FM_NUMERIC_ON=side timeout 300 python bench.py --steps 50 --warmup 10 --no-extra-configs --no-cpu-baseline > $out/bench_fm_numeric_on_side.json 2>/dev/null; ms fm_numeric_on_side
B="--steps 50 --warmup 10 --no-extra-configs --no-cpu-baseline"
RBX_FM_TIER_C=1 timeout 300 python bench.py $B > $out/bench_fm_tier_c.json 2>/dev/null; ms fm_tier_c
RBX_FM_TIER_C=1 RECBOX_AMD_FM_BLOCKSORT_AT=side timeout 300 python bench.py $B > $out/bench_fm_tier_c_blocksort_side.json 2>/dev/null; ms fm_tier_c_blocksort_side
RBX_FM_TIER_C=1 RECBOX_AMD_FM_BLOCKSORT_AT=side RECBOX_AMD_FM_REZERO_ON=fused timeout 300 python bench.py $B > $out/bench_fm_tier_c_fused_clear.json 2>/dev/null; ms fm_tier_c_fused_clear
timeout 300 python bench.py $B --dist zipf > $out/bench_fm_zipf.json 2>/dev/null; ms fm_zipf
RBX_FM_TIER_C=1 RECBOX_AMD_FM_BLOCKSORT_AT=side timeout 300 python bench.py $B --dist zipf > $out/bench_fm_zipf_tier_c.json 2>/dev/null; ms fm_zipf_tier_c
timeout 300 python bench.py $B --fresh-grads > $out/bench_fm_fresh_grads.json 2>/dev/null; ms fm_fresh_grads
RECBOX_BENCH_ONE_GPU=1 timeout 600 python bench.py --gpus 2 --scaling strong --steps 10 --warmup 3 > $out/bench_fm_two_ranks_one_gpu_strong.json 2>/dev/null; ms fm_two_ranks_one_gpu_strong
prof() { # name, env, bench args, anchor kernel, occurrence
  rm -rf $out/prof
  (cd /tmp && env $2 timeout 600 rocprofv3 --kernel-trace --stats -d $out/prof -o b -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-extra-configs $3 > $out/prof_$1.log 2>&1)
  db=$(find $out/prof -name "*.db" | head -1)
  python profiles/topk.py $db 25 > $out/$1_kernel_stats.txt
  python profiles/timeline.py $db "$4" $5 > $out/$1_replay_timeline.txt 2>&1
  if [ $1 = fm ]; then python profiles/kernel_slice.py $db fm_fused_fwd 24 60 > $out/fm_fwd_kernel_by_phase.txt 2>&1; fi
  rm -rf $out/prof $out/prof_$1.log
}
prof fm "X=1" "--steps 20 --warmup 5" compact_ids 30
prof fm_tier_c "RBX_FM_TIER_C=1 RECBOX_AMD_FM_BLOCKSORT_AT=side" "--steps 20 --warmup 5" compact_ids 30
prof fm_sharded1 "X=1" "--config fm --force-sharded --steps 20 --warmup 5" route_count 14
prof youtubednn "X=1" "--config youtubednn --steps 20 --warmup 5" embed_seq 12
prof deepfm "X=1" "--config deepfm --steps 20 --warmup 5" deepfm 12
prof sasrec "X=1" "--config sasrec --steps 20 --warmup 5" embed_seq 12
prof sasrec_seqblock_off "RECBOX_AMD_SEQBLOCK=0" "--config sasrec --steps 20 --warmup 5" embed_seq 12
prof youtubednn_sharded1 "X=1" "--config youtubednn --force-sharded --steps 20 --warmup 5" "shard_count_kernel<true>" 14
prof deepfm_sharded1 "X=1" "--config deepfm --force-sharded --steps 20 --warmup 5" "shard_count_kernel<false>" 14
# HBM traffic of the FM kernels from the PMC counters, separate passes, tier C off (the default) and on
for tc in 0 1; do
  for c in FETCH_SIZE WRITE_SIZE; do
    (cd /tmp && RBX_FM_TIER_C=$tc timeout 400 rocprofv3 --kernel-trace --pmc $c -d $out/pmc_fm_$c -o b -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-extra-configs --eager --steps 5 --warmup 3 > /dev/null 2>&1)
    python profiles/pmc.py $(find $out/pmc_fm_$c -name "*.db" | head -1) $c > $out/pmc_fm_tierc${tc}_$c.txt
    rm -rf $out/pmc_fm_$c
  done
done
# matrix-pipe utilisation of SASRec's kernels (one PMC pass, eager launches)
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE -d $out/pmc_mfma_sasrec -o b -- python $GRAFT_REPO_ROOT/bench.py --config sasrec --eager --steps 3 --warmup 2 --no-cpu-baseline > /dev/null 2>&1)
timeout 120 python profiles/mfma_util.py $(find $out/pmc_mfma_sasrec -name "*.db" | head -1) attn gemm tall_dw sb_ < /dev/null > $out/mfma_util_sasrec.txt 2>&1
rm -rf $out/pmc_mfma_sasrec
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1 | tee -a $out/summary.txt
