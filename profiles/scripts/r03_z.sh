cd /root/repo; mkdir -p gpurun_out
export TMPDIR=/tmp
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES -d /tmp/sq -o b -- python /root/repo/bench.py --config sasrec --eager --steps 3 --warmup 2 --no-cpu-baseline > /dev/null 2>&1)
python profiles/sq_stalls.py $(find /tmp/sq -name "*.db" | head -1) > gpurun_out/sq_stalls_sasrec.txt 2>&1
rm -rf /tmp/sq
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_INST_CYCLES_VMEM SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_FLAT SQ_ACTIVE_INST_VMEM -d /tmp/sq2 -o b -- python /root/repo/bench.py --config sasrec --eager --steps 3 --warmup 2 --no-cpu-baseline > /tmp/sq2.log 2>&1)
python profiles/pmc.py $(find /tmp/sq2 -name "*.db" | head -1) SQ_ACTIVE_INST_VALU 2>&1 | grep -i attn > gpurun_out/sq2.txt
for c in SQ_WAVE_CYCLES SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM; do python profiles/pmc.py $(find /tmp/sq2 -name "*.db" | head -1) $c 2>&1 | grep -i attn >> gpurun_out/sq2.txt; done
tail -5 /tmp/sq2.log >> gpurun_out/sq2.txt
