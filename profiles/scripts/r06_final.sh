#!/bin/bash
# round 6 evidence, ONE call on one box: full GPU suite (parity ledger), the driver's bench line, world-of-one sharded lines,
# FM variants (Zipf, fresh gradients, quad forward off), rocprofv3 kernel stats + replay timelines of every config, PMC traffic
# of the FM and YoutubeDNN kernels (separate passes), matrix-pipe utilisation of SASRec and DeepFM, smoke
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/r06
rm -rf $out; mkdir -p $out
rm -f gpurun_out/parity_errors.txt
timeout 2400 python -m pytest tests -q -m gpu > $out/gpu_tests.log 2>&1
echo "gpu tests exit $?" | tee -a $out/summary.txt; tail -3 $out/gpu_tests.log | tee -a $out/summary.txt
cp gpurun_out/parity_errors.txt $out/parity_errors.txt 2>/dev/null
ms() { python -c "import json,sys; d=json.loads([l for l in open('$out/bench_$1.json') if l.startswith('{')][-1]); r=d.get('roofline') or {}; print('$1', round(d['ms_per_step'],4), r.get('kernel'), r.get('kernel_ms'), r.get('frac'))" 2>&1 | tail -1 | tee -a $out/summary.txt; }
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $out/bench_fm.json 2> $out/bench_fm.err; ms fm
for cfg in youtubednn deepfm sasrec; do
  timeout 600 python bench.py --config $cfg --steps 20 --warmup 5 > $out/bench_$cfg.json 2>/dev/null; ms $cfg
done
for cfg in fm youtubednn deepfm; do
  timeout 300 python bench.py --config $cfg --force-sharded --steps 30 --warmup 5 --no-cpu-baseline > $out/bench_${cfg}_sharded1.json 2> $out/bench_${cfg}_sharded1.err; ms ${cfg}_sharded1
done
B="--steps 48 --warmup 8 --no-extra-configs --no-cpu-baseline"
timeout 300 python bench.py $B > $out/bench_fm_again.json 2>/dev/null; ms fm_again
timeout 300 python bench.py $B --steps-per-graph 1 > $out/bench_fm_one_step_per_graph.json 2>/dev/null; ms fm_one_step_per_graph
timeout 300 python bench.py $B --steps-per-graph 1 --dist zipf > $out/bench_fm_zipf_one_step_per_graph.json 2>/dev/null; ms fm_zipf_one_step_per_graph
RBX_FM_QUAD=0 timeout 300 python bench.py $B > $out/bench_fm_quad_off.json 2>/dev/null; ms fm_quad_off
timeout 300 python bench.py $B --dist zipf > $out/bench_fm_zipf.json 2>/dev/null; ms fm_zipf
timeout 300 python bench.py $B --fresh-grads > $out/bench_fm_fresh_grads.json 2>/dev/null; ms fm_fresh_grads
RECBOX_BENCH_ONE_GPU=1 timeout 600 python bench.py --gpus 2 --scaling strong --steps 10 --warmup 3 > $out/bench_fm_two_ranks_one_gpu_strong.json 2>/dev/null; ms fm_two_ranks_one_gpu_strong
prof() { # name, bench args, anchor kernel, occurrence
  rm -rf $out/prof
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $out/prof -o b -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-extra-configs $2 > $out/prof_$1.log 2>&1)
  db=$(find $out/prof -name "*.db" | head -1)
  python profiles/topk.py $db 30 > $out/$1_kernel_stats.txt
  python profiles/timeline.py $db "$3" $4 $5 > $out/$1_replay_timeline.txt 2>&1
  if [ $1 = fm ]; then python profiles/kernel_slice.py $db fm_quad_fwd 24 60 > $out/fm_fwd_kernel_by_phase.txt 2>&1; fi
  rm -rf $out/prof $out/prof_$1.log
}
prof fm "--steps 20 --warmup 5" rezero_rows 26 5
mv $out/fm_kernel_stats.txt $out/fm_kernel_stats_keep.txt
prof fm_one_step_per_graph "--steps 20 --warmup 5 --steps-per-graph 1" rezero_rows 28 2
mv $out/fm_kernel_stats_keep.txt $out/fm_kernel_stats.txt
prof fm_sharded1 "--config fm --force-sharded --steps 20 --warmup 5" route_count 14
prof youtubednn "--config youtubednn --steps 20 --warmup 5" embed_seq 35
prof deepfm "--config deepfm --steps 20 --warmup 5" "embed_fwd_kernel<16" 35
prof sasrec "--config sasrec --steps 20 --warmup 5" embed_seq 35
# HBM traffic from the PMC counters, separate passes (FETCH_SIZE x 2 + WRITE_SIZE, MI355X_MICROARCH.md)
for cfg in fm youtubednn; do
  for c in FETCH_SIZE WRITE_SIZE; do
    extra="--no-extra-configs"; [ $cfg != fm ] && extra="--config $cfg"
    (cd /tmp && timeout 500 rocprofv3 --kernel-trace --pmc $c -d $out/pmc_${cfg}_$c -o b -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline $extra --eager --steps 5 --warmup 3 > /dev/null 2>&1)
    python profiles/pmc.py $(find $out/pmc_${cfg}_$c -name "*.db" | head -1) $c > $out/pmc_${cfg}_$c.txt
    rm -rf $out/pmc_${cfg}_$c
  done
done
# matrix-pipe utilisation (one PMC pass each, eager launches)
for cfg in sasrec deepfm; do
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE -d $out/pmc_mfma_$cfg -o b -- python $GRAFT_REPO_ROOT/bench.py --config $cfg --eager --steps 3 --warmup 2 --no-cpu-baseline > /dev/null 2>&1)
  timeout 120 python profiles/mfma_util.py $(find $out/pmc_mfma_$cfg -name "*.db" | head -1) attn gemm tall_dw sb_ < /dev/null > $out/mfma_util_$cfg.txt 2>&1
  rm -rf $out/pmc_mfma_$cfg
done
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1 | tee -a $out/summary.txt
cat $out/summary.txt
