#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/r04l
mkdir -p $out
ms() { python -c "import json,sys; d=json.loads([l for l in open('$out/bench_$1.json') if l.startswith('{')][-1]); r=d.get('roofline') or {}; print('$1', round(d['ms_per_step'],4), round(r.get('kernel_ms')*1e3,1), round(r.get('frac'),3))" 2>&1 | tail -1 | tee -a $out/summary.txt; }
B="--steps 50 --warmup 10 --no-extra-configs --no-cpu-baseline"
timeout 300 python bench.py $B > $out/bench_default.json 2>/dev/null; ms default
RBX_FM_TIER_C=1 RECBOX_AMD_FM_BLOCKSORT_AT=side timeout 300 python bench.py $B > $out/bench_tc_side.json 2>/dev/null; ms tc_side
RBX_FM_TIER_C=1 timeout 300 python bench.py $B --sort-after-forward > $out/bench_tc_after.json 2>/dev/null; ms tc_after
RBX_FM_TIER_C=1 RECBOX_AMD_FM_BLOCKSORT_AT=side timeout 300 python bench.py $B --sort-after-forward > $out/bench_tc_after_bside.json 2>/dev/null; ms tc_after_bside
RBX_FM_TIER_C=1 RECBOX_AMD_FM_IDS_WORK=side timeout 300 python bench.py $B --sort-after-forward > $out/bench_tc_after_idsside.json 2>/dev/null; ms tc_after_idsside
timeout 300 python bench.py $B --sort-after-forward > $out/bench_default_after.json 2>/dev/null; ms default_after
timeout 300 python bench.py $B > $out/bench_default2.json 2>/dev/null; ms default2
rm -rf $out/prof
(cd /tmp && RBX_FM_TIER_C=1 RECBOX_AMD_FM_BLOCKSORT_AT=side timeout 600 rocprofv3 --kernel-trace --stats -d $out/prof -o b -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-extra-configs --steps 20 --warmup 5 --sort-after-forward > $out/prof.log 2>&1)
db=$(find $out/prof -name "*.db" | head -1)
python profiles/timeline.py $db compact_ids 30 > $out/tc_after_bside_timeline.txt 2>&1
rm -rf $out/prof
