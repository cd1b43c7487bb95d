#!/bin/bash
# round 5: embed_seq_kernel rows in flight per lane (4 = default, 6, 8), alternated three times: the forward alone at cfg 3's
# shape and the YoutubeDNN bench line
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05z2
mkdir -p $O
export TMPDIR=/tmp
for rep in 1 2 3; do
for v in default seq_u6 seq_u8 seq_u8w8; do
  if [ $v != default ]; then export RECBOX_HIP_LIB=$GRAFT_REPO_ROOT/recbox_amd/lib/librecbox_hip_$v.so; else unset RECBOX_HIP_LIB; fi
  timeout 200 python profiles/ubench/seq_gather_lab.py 2>&1 | grep "us " | tee -a $O/seq_gather_lab.txt
  timeout 300 python bench.py --config youtubednn --steps 30 --warmup 5 --no-cpu-baseline > $O/b.json 2> $O/b.err
  python -c "
import json
d=json.loads([l for l in open('$O/b.json') if l.startswith('{')][-1]); r=d['roofline']
print('%-12s youtubednn ms_per_step %.4f  gather %.1f us  frac %.3f' % ('$v', d['ms_per_step'], r['kernel_ms']*1e3, r['frac']))" | tee -a $O/seq_gather_lab.txt
done
done
