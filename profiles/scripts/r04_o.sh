#!/bin/bash
# PMC passes over the attention kernels alone (profiles/ubench/attn_stream): wavefront stall split and MFMA pipe utilisation
out=/root/repo/gpurun_out/r04o
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
BWD=${BWD:-0}
for S in 0 1; do
  (RBX_ATTN_STREAM=$S timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES -d $out/pmc_sq_$S -o b -- /root/repo/profiles/ubench/attn_stream 200 4096 $BWD > $out/pmc_sq_$S.log 2>&1)
  python /root/repo/profiles/sq_stalls.py $(find $out/pmc_sq_$S -name "*.db" | head -1) attn > $out/sq_stalls_stream$S.txt 2>&1
  rm -rf $out/pmc_sq_$S
  (RBX_ATTN_STREAM=$S timeout 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE -d $out/pmc_mfma_$S -o b -- /root/repo/profiles/ubench/attn_stream 200 4096 $BWD > $out/pmc_mfma_$S.log 2>&1)
  python /root/repo/profiles/mfma_util.py $(find $out/pmc_mfma_$S -name "*.db" | head -1) attn > $out/mfma_util_stream$S.txt 2>&1
  rm -rf $out/pmc_mfma_$S
done
cd /root/repo
for S in 0 1; do RBX_ATTN_STREAM=$S timeout 120 profiles/ubench/attn_stream 200 4096 $BWD | grep -v launch; done > $out/attn_stream.txt 2>&1
cat $out/*.txt
