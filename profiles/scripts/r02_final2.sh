#!/bin/bash
# after r02_final.sh: MFMA utilisation of the rewritten GEMM loop (cfg 4) and the per-shape GEMM table, new loop vs -DRBX_GEMM_PIPE=0
out=/root/repo/gpurun_out/r2final
mkdir -p $out
export TMPDIR=/tmp PYTHONPATH=/root/repo
cd /root/repo
for cfg in deepfm sasrec; do
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE -d $out/pmc_mfma_$cfg -o b -- python /root/repo/bench.py --config $cfg --eager --steps 3 --warmup 2 --no-cpu-baseline > $out/pmc_mfma_$cfg.log 2>&1)
  python profiles/mfma_util.py $(find $out/pmc_mfma_$cfg -name "*.db" | head -1) > $out/mfma_util_$cfg.txt 2>&1
  rm -rf $out/pmc_mfma_$cfg
  head -12 $out/mfma_util_$cfg.txt
done
(timeout 300 python profiles/gemm_shapes.py; timeout 300 python profiles/gemm_shapes.py --edges) 2>&1 | grep -v amdgpu.ids > $out/gemm_shapes_new.txt
if [ -f recbox_amd/lib/variants/pipe0.so ]; then  # (only when that variant library has been built: profiles/scripts/build_variant.sh pipe0 -DRBX_GEMM_PIPE=0)
  (RECBOX_HIP_LIB=recbox_amd/lib/variants/pipe0.so timeout 300 python profiles/gemm_shapes.py; RECBOX_HIP_LIB=recbox_amd/lib/variants/pipe0.so timeout 300 python profiles/gemm_shapes.py --edges) 2>&1 | grep -v amdgpu.ids > $out/gemm_shapes_tested_loop_only.txt
fi
cat $out/gemm_shapes_new.txt
