#!/bin/bash
# GEMM k loop: branch-free steady state with the LDS stores in the MFMA shadow (RBX_GEMM_PIPE), wave priority in the MFMA block
out=/root/repo/gpurun_out/r2gp
rm -rf $out; mkdir -p $out
export TMPDIR=/tmp PYTHONPATH=/root/repo
cd /root/repo
timeout 900 python -m pytest tests/test_gpu_matching.py tests/test_gpu_cabi_vs_c_oracle.py -x -q -m gpu -k "linear or mlp or dense or gemm or tower or deepfm or dssm" 2>&1 | tail -3
for v in "" pipe0 prio1; do
  echo "== variant ${v:-default}"
  if [ -n "$v" ]; then export RECBOX_HIP_LIB=recbox_amd/lib/variants/$v.so; fi
  timeout 300 python profiles/gemm_shapes.py 2>&1 | grep -v amdgpu.ids | tee $out/shapes_${v:-default}.txt
done
