#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/r04m
mkdir -p $out
ms() { python -c "import json,sys; d=json.loads([l for l in open('$out/bench_$1.json') if l.startswith('{')][-1]); r=d.get('roofline') or {}; print('$1', round(d['ms_per_step'],4), round(r.get('kernel_ms')*1e3,1), round(r.get('frac'),3), round(r.get('frac_alone',0),3))" 2>&1 | tail -1 | tee -a $out/summary.txt; }
B="--steps 50 --warmup 10 --no-extra-configs --no-cpu-baseline"
timeout 300 python bench.py $B > $out/bench_default.json 2>/dev/null; ms default
timeout 300 python bench.py $B --pack-tables > $out/bench_pack.json 2>$out/pack.err; ms pack
timeout 300 python bench.py $B --pack-tables --pack-min-vocab 90000 > $out/bench_pack_big.json 2>/dev/null; ms pack_big
timeout 300 python bench.py $B --contiguous-ids > $out/bench_contig.json 2>/dev/null; ms contig
timeout 300 python bench.py $B > $out/bench_default2.json 2>/dev/null; ms default2
