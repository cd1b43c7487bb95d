#!/bin/bash
export TMPDIR=/tmp PYTHONPATH=/root/repo
cd /root/repo
timeout 900 python -m pytest tests/test_gpu_matching.py tests/test_gpu_cabi_vs_c_oracle.py tests/test_gpu_edge_cases.py -x -q -m gpu 2>&1 | tail -3
for v in "" ; do
  echo "== variant ${v:-default}"
  if [ -n "$v" ]; then export RECBOX_HIP_LIB=recbox_amd/lib/variants/$v.so; fi
  timeout 300 python profiles/gemm_shapes.py 2>&1 | grep -v amdgpu.ids
  timeout 300 python profiles/gemm_shapes.py --edges 2>&1 | grep -v amdgpu.ids
done
