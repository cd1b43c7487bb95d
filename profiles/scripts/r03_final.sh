#!/bin/bash
# end-of-round evidence (round 3): full GPU suite, the default bench line (FM + the three model configs under "configs"),
# world-of-one sharded forms, FM variants, optimiser lines, rocprofv3 kernel stats of the same commands, one replay timeline
# of the FM step, the pieces of the FM step alone, PMC traffic of the FM kernels
out=/root/repo/gpurun_out/r3final
rm -rf $out; mkdir -p $out
export TMPDIR=/tmp
cd /root/repo
timeout 2400 python -m pytest tests -x -q -m gpu > $out/gpu_tests.log 2>&1
tail -3 $out/gpu_tests.log
timeout 900 python bench.py > $out/bench_fm.json 2>$out/bench_fm.err; cut -c1-400 $out/bench_fm.json
python -c "
import json
d=json.loads(open('$out/bench_fm.json').readline())
for k,v in d.get('configs',{}).items():
    open('$out/bench_%s.json'%k,'w').write(json.dumps(v)+'\n')
    print(k, v.get('ms_per_step'), (v.get('roofline') or {}).get('frac'), v.get('skipped'))"
ms() { python -c "
import json
d=json.loads(open('$out/bench_$1.json').readline()); print('$1', round(d['ms_per_step'],4))"; }
timeout 600 python bench.py --config youtubednn --force-sharded --no-cpu-baseline > $out/bench_youtubednn_sharded1.json 2>/dev/null; ms youtubednn_sharded1
timeout 600 python bench.py --config deepfm --force-sharded --no-cpu-baseline > $out/bench_deepfm_sharded1.json 2>/dev/null; ms deepfm_sharded1
timeout 600 python bench.py --force-sharded --no-cpu-baseline > $out/bench_fm_sharded1.json 2>/dev/null; ms fm_sharded1
timeout 600 python bench.py --rotate-by-copy --no-cpu-baseline > $out/bench_fm_rotate_by_copy.json 2>/dev/null; ms fm_rotate_by_copy
timeout 600 python bench.py --rotate 1 --no-cpu-baseline > $out/bench_fm_onebatch.json 2>/dev/null; ms fm_onebatch
timeout 600 python bench.py --dist zipf --no-cpu-baseline > $out/bench_fm_zipf.json 2>/dev/null; ms fm_zipf
RBX_FM_TIER_A=0 timeout 600 python bench.py --no-cpu-baseline > $out/bench_fm_one_tier.json 2>/dev/null; ms fm_one_tier
RBX_FM_TIER_A_VMAX=16384 timeout 600 python bench.py --no-cpu-baseline > $out/bench_fm_vmax16384.json 2>/dev/null; ms fm_vmax16384
RECBOX_AMD_FM_REZERO_ON=side timeout 600 python bench.py --no-cpu-baseline > $out/bench_fm_rezero_beside.json 2>/dev/null; ms fm_rezero_beside
RECBOX_AMD_FM_TWO_CHAINS=0 timeout 600 python bench.py --no-cpu-baseline > $out/bench_fm_one_chain.json 2>/dev/null; ms fm_one_chain
# SASRec / DeepFM: what this round's kernels are worth, one switch at a time
RBX_GEMM_STREAM64=0 timeout 600 python bench.py --config sasrec --no-cpu-baseline --steps 20 --warmup 5 > $out/bench_sasrec_tile_gemm.json 2>/dev/null; ms sasrec_tile_gemm
RECBOX_AMD_SHARE_TABLE_GRADS=0 timeout 600 python bench.py --config sasrec --no-cpu-baseline --steps 20 --warmup 5 > $out/bench_sasrec_two_grads.json 2>/dev/null; ms sasrec_two_grads
RBX_ATTN_SPLIT=0 timeout 600 python bench.py --config sasrec --no-cpu-baseline --steps 20 --warmup 5 > $out/bench_sasrec_one_wave_per_tile.json 2>/dev/null; ms sasrec_one_wave_per_tile
RECBOX_AMD_FFN_MASK_IN_GEMMS=0 timeout 600 python bench.py --config sasrec --no-cpu-baseline --steps 20 --warmup 5 > $out/bench_sasrec_ffn_mask_pass.json 2>/dev/null; ms sasrec_ffn_mask_pass
RECBOX_AMD_SEQ_POSITIONS=0 timeout 600 python bench.py --config sasrec --no-cpu-baseline --steps 20 --warmup 5 > $out/bench_sasrec_position_lookup.json 2>/dev/null; ms sasrec_position_lookup
RECBOX_AMD_BN_IN_GEMM=fwd timeout 600 python bench.py --config deepfm --no-cpu-baseline --steps 20 --warmup 5 > $out/bench_deepfm_bn_fwd_in_gemm.json 2>/dev/null; ms deepfm_bn_fwd_in_gemm
RECBOX_AMD_BN_IN_GEMM=1 timeout 600 python bench.py --config deepfm --no-cpu-baseline --steps 20 --warmup 5 > $out/bench_deepfm_bn_in_gemm.json 2>/dev/null; ms deepfm_bn_in_gemm
RECBOX_AMD_GEMM_BX6=0 RBX_GEMM_BX6=0 timeout 600 python bench.py --config deepfm --no-cpu-baseline --steps 20 --warmup 5 > $out/bench_deepfm_f32_mfma.json 2>/dev/null; ms deepfm_f32_mfma
RBX_GEMM_BX6_DW=0 timeout 600 python bench.py --config deepfm --no-cpu-baseline --steps 20 --warmup 5 > $out/bench_deepfm_dw_f32.json 2>/dev/null; ms deepfm_dw_f32
RBX_GEMM_BX6=2 timeout 600 python bench.py --config deepfm --no-cpu-baseline --steps 20 --warmup 5 > $out/bench_deepfm_bx6_128.json 2>/dev/null; ms deepfm_bx6_128
RECBOX_AMD_GEMM_BX6=0 RBX_GEMM_BX6=0 timeout 600 python bench.py --config youtubednn --no-cpu-baseline --steps 20 --warmup 5 > $out/bench_youtubednn_f32_mfma.json 2>/dev/null; ms youtubednn_f32_mfma
for v in 1 0; do
  RECBOX_AMD_GEMM_BX6=$v RBX_GEMM_BX6=$v PYTHONPATH=/root/repo timeout 300 python profiles/gemm_shapes.py 2>&1 | grep -v amdgpu.ids > $out/gemm_shapes_bx6_$v.txt
done
for o in sparse_adam dense_adam; do
  timeout 900 python bench.py --config youtubednn --no-cpu-baseline --optimizer $o --steps 20 --warmup 5 > $out/bench_youtubednn_$o.json 2>/dev/null; ms youtubednn_$o
  timeout 900 python bench.py --config deepfm --no-cpu-baseline --optimizer $o --steps 20 --warmup 5 > $out/bench_deepfm_$o.json 2>/dev/null; ms deepfm_$o
done
prof() { # name, bench args
  rm -rf $out/prof
  (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d $out/prof -o b -- python /root/repo/bench.py --no-cpu-baseline $2 > $out/prof_$1.log 2>&1)
  db=$(find $out/prof -name "*.db" | head -1)
  python profiles/topk.py $db 48 > $out/$1_kernel_stats.txt
  if [ $1 = fm ]; then python profiles/timeline.py $db compact_ids 30 > $out/fm_replay_timeline.txt 2>&1; fi
  if [ $1 = fm ]; then python profiles/kernel_slice.py $db fm_fused_fwd 24 60 > $out/fm_fwd_kernel_by_phase.txt 2>&1; fi
  rm -rf $out/prof
}
prof fm ""
prof youtubednn "--config youtubednn --steps 20 --warmup 5"
prof deepfm "--config deepfm --steps 20 --warmup 5"
prof sasrec "--config sasrec --steps 20 --warmup 5"
prof youtubednn_sharded1 "--config youtubednn --force-sharded --steps 20 --warmup 5"
# MFMA pipe utilisation of the tower / attention kernels
for cfg in deepfm sasrec; do
  (cd /tmp && timeout 400 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE -d $out/pmc_mfma_$cfg -o b -- python /root/repo/bench.py --config $cfg --eager --steps 3 --warmup 2 --no-cpu-baseline > /dev/null 2>&1)
  python profiles/mfma_util.py $(find $out/pmc_mfma_$cfg -name "*.db" | head -1) > $out/mfma_util_$cfg.txt 2>&1
  rm -rf $out/pmc_mfma_$cfg
done
# the pieces of the FM step, each alone
python profiles/ubench/fm_bwd_parts.py 20 2>&1 | grep -v "Warn\|amdgpu.ids" > $out/fm_step_pieces_alone.txt
rm -rf $out/prof
(cd /tmp && rocprofv3 --kernel-trace --stats -d $out/prof -o b -- python /root/repo/profiles/ubench/fm_bwd_parts.py 20 > /dev/null 2>&1)
python profiles/topk.py $(find $out/prof -name "*.db" | head -1) >> $out/fm_step_pieces_alone.txt
rm -rf $out/prof
# HBM traffic of the FM kernels from the PMC counters, separate passes
for c in FETCH_SIZE WRITE_SIZE; do
  (cd /tmp && timeout 400 rocprofv3 --kernel-trace --pmc $c -d $out/pmc_fm_$c -o b -- python /root/repo/bench.py --no-cpu-baseline --eager --steps 5 --warmup 3 > /dev/null 2>&1)
  python profiles/pmc.py $(find $out/pmc_fm_$c -name "*.db" | head -1) $c > $out/pmc_fm_$c.txt
  rm -rf $out/pmc_fm_$c
done
grep -h -E "fm_fused_fwd|segment_reduce|ta_reduce|ta_final|rezero" $out/pmc_*.txt | cut -c1-160
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
