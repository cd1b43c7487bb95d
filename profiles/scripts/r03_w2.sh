#!/bin/bash
out=/root/repo/gpurun_out/r3w
mkdir -p $out
export TMPDIR=/tmp
cd /root/repo
(cd /tmp && PYTHONPATH=/root/repo timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES -d $out/pmc1 -o b -- python /root/repo/profiles/gemm_shapes.py > $out/pmc1.log 2>&1)
python profiles/sq_stalls.py $(find $out/pmc1 -name "*.db" | head -1) gemm > $out/sq_stalls.txt 2>&1
rm -rf $out/pmc1
cat $out/sq_stalls.txt
