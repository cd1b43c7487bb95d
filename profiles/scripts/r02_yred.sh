#!/bin/bash
export TMPDIR=/tmp PYTHONPATH=/root/repo
cd /root/repo
for v in "" ck24 ck64 rw4 rw6; do
  echo "== ${v:-default}"
  if [ -n "$v" ]; then export RECBOX_HIP_LIB=recbox_amd/lib/variants/$v.so; fi
  timeout 400 python bench.py --config youtubednn --no-cpu-baseline 2>/dev/null | grep -o '"ms_per_step": [0-9.]*'
  timeout 400 python bench.py --config deepfm --no-cpu-baseline 2>/dev/null | grep -o '"ms_per_step": [0-9.]*'
  timeout 400 python bench.py --no-cpu-baseline 2>/dev/null | grep -o '"ms_per_step": [0-9.]*'
done
