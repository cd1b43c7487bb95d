#!/bin/bash
out=/root/repo/gpurun_out/r3v
rm -rf $out; mkdir -p $out
cd /root/repo
for v in "" nt00 nt10 nt01; do
  for i in 1 2; do
  if [ -n "$v" ]; then export RECBOX_HIP_LIB=recbox_amd/lib/variants/$v.so; else unset RECBOX_HIP_LIB; fi
  timeout 600 python bench.py --config sasrec --no-cpu-baseline --steps 20 --warmup 5 > $out/b.json 2>$out/err.txt
  python -c "
import json
d=json.loads(open('$out/b.json').readline()); print('sasrec [$v]', round(d['ms_per_step'],4))" || tail -3 $out/err.txt
  done
done
