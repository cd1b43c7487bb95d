#!/bin/bash
# gather with pooled history (embed_seq_kernel, D = 128): rows in flight per lane / wavefronts per SIMD
export TMPDIR=/tmp PYTHONPATH=/root/repo
cd /root/repo
for v in "" su8 su2 sw6 sw8; do
  echo "== ${v:-default}"
  if [ -n "$v" ]; then export RECBOX_HIP_LIB=recbox_amd/lib/variants/$v.so; fi
  timeout 400 python bench.py --config youtubednn --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d=json.loads(sys.stdin.readline()); print('ms_per_step', round(d['ms_per_step'],4), 'gather_ms', round(d['roofline']['kernel_ms'],4), 'frac', round(d['roofline']['frac'],3))"
done
