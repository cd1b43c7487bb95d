#!/bin/bash
# round 5: the FM step's existing placement switches re-measured with the quad forward (one box, back to back)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05f
mkdir -p $O
run() {  # name, env...
  n=$1; shift
  env "${@:1:$#}" timeout 200 python bench.py --steps 50 --warmup 10 --no-extra-configs --no-cpu-baseline $BENCH_EXTRA > $O/bench_$n.json 2> $O/bench_$n.err
  python - <<PY
import json
try:
    d=json.loads([l for l in open('$O/bench_$n.json') if l.startswith('{')][-1]); r=d['roofline']
    print('%-28s ms_per_step %.4f  fwd %.1f us (alone %.1f)' % ('$n', d['ms_per_step'], r['kernel_ms']*1e3, (r.get('kernel_ms_alone') or 0)*1e3))
except Exception as e:
    print('$n', 'failed', e)
PY
}
run default A=1
run default_again A=1
run ids_side RECBOX_AMD_FM_IDS_WORK=side
run blocksort_side RECBOX_AMD_FM_BLOCKSORT_AT=side
run ids_side_blocksort_side RECBOX_AMD_FM_IDS_WORK=side RECBOX_AMD_FM_BLOCKSORT_AT=side
run tier_c RBX_FM_TIER_C=1
run tier_c_blocksort_side RBX_FM_TIER_C=1 RECBOX_AMD_FM_BLOCKSORT_AT=side
run one_chain RECBOX_AMD_FM_TWO_CHAINS=0
run sort_after_fwd RECBOX_AMD_SORT_FIRST=0
BENCH_EXTRA="--dist zipf" run zipf A=1
BENCH_EXTRA="--dist zipf" run zipf_tier_c RBX_FM_TIER_C=1 RECBOX_AMD_FM_BLOCKSORT_AT=side
