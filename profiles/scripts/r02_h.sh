#!/bin/bash
# MFMA utilisation counters (DeepFM towers, SASRec attention) + HBM traffic of the gathers (FM fwd, cfg-3 gather / serve)
out=/root/repo/gpurun_out/r2h
rm -rf $out; mkdir -p $out
export TMPDIR=/tmp
cd /root/repo
timeout 600 python -m pytest tests/test_gpu_matching.py -x -q -m gpu -k "batch_norm or epilogue" > $out/tests.log 2>&1; tail -3 $out/tests.log
rocprofv3 -L 2>/dev/null | grep -i -E "MFMA|SQ_BUSY_CYCLES|GRBM_GUI_ACTIVE" | head -30 > $out/counters_available.txt
for cfg in deepfm sasrec; do
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE -d $out/pmc_mfma_$cfg -o b -- python /root/repo/bench.py --config $cfg --eager --steps 3 --warmup 2 --no-cpu-baseline > $out/pmc_mfma_$cfg.log 2>&1)
  python profiles/mfma_util.py $(find $out/pmc_mfma_$cfg -name "*.db" | head -1) > $out/mfma_util_$cfg.txt 2>&1
  rm -rf $out/pmc_mfma_$cfg
  cat $out/mfma_util_$cfg.txt | head -14
done
for c in FETCH_SIZE WRITE_SIZE; do
  (cd /tmp && timeout 400 rocprofv3 --kernel-trace --pmc $c -d $out/pmc_fm_$c -o b -- python /root/repo/bench.py --no-cpu-baseline --eager --steps 5 --warmup 3 > /dev/null 2>&1)
  python profiles/pmc.py $(find $out/pmc_fm_$c -name "*.db" | head -1) $c > $out/pmc_fm_$c.txt
  rm -rf $out/pmc_fm_$c
  (cd /tmp && timeout 400 rocprofv3 --kernel-trace --pmc $c -d $out/pmc_yt_$c -o b -- python /root/repo/bench.py --config youtubednn --no-cpu-baseline --eager --steps 3 --warmup 2 > /dev/null 2>&1)
  python profiles/pmc.py $(find $out/pmc_yt_$c -name "*.db" | head -1) $c > $out/pmc_yt_$c.txt
  rm -rf $out/pmc_yt_$c
  (cd /tmp && timeout 400 rocprofv3 --kernel-trace --pmc $c -d $out/pmc_yts_$c -o b -- python /root/repo/bench.py --config youtubednn --force-sharded --no-cpu-baseline --steps 3 --warmup 2 > /dev/null 2>&1)
  python profiles/pmc.py $(find $out/pmc_yts_$c -name "*.db" | head -1) $c > $out/pmc_yts_$c.txt
  rm -rf $out/pmc_yts_$c
done
grep -h -E "fm_fused_fwd|segment_reduce|embed_seq|shard_serve|embed_fwd" $out/pmc_*.txt | cut -c1-160
