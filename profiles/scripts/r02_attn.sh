#!/bin/bash
export TMPDIR=/tmp PYTHONPATH=/root/repo
cd /root/repo
timeout 900 python -m pytest tests/test_gpu_matching.py tests/test_gpu_cabi_vs_c_oracle.py tests/test_gpu_edge_cases.py -x -q -m gpu -k "attn or attention or sasrec or mha" 2>&1 | tail -3
timeout 400 python bench.py --config sasrec --no-cpu-baseline 2>/dev/null | cut -c1-900
