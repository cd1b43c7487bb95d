#!/bin/bash
out=/root/repo/gpurun_out/r2p
rm -rf $out; mkdir -p $out
export TMPDIR=/tmp
cd /root/repo
for cfg in deepfm sasrec; do
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES -d $out/pmc_$cfg -o b -- python /root/repo/bench.py --config $cfg --eager --steps 3 --warmup 2 --no-cpu-baseline > $out/pmc_$cfg.log 2>&1)
  python profiles/sq_stalls.py $(find $out/pmc_$cfg -name "*.db" | head -1) gemm attn tall_dw > $out/sq_stalls_$cfg.txt 2>&1
  rm -rf $out/pmc_$cfg
  cat $out/sq_stalls_$cfg.txt
done
