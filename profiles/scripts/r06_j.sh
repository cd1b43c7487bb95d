#!/bin/bash
# round 6: this tree against round 5's last commit (worktree _r5, built here) on ONE box: the sharded FM step in a world of one and the other bench lines
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r06j
mkdir -p $O
export TMPDIR=/tmp
line() { python - "$1" "$2" <<'PY'
import json, sys
txt = [l for l in open(sys.argv[1]).read().splitlines() if l.startswith("{")]
d = json.loads(txt[-1])
print("%s ms_per_step %.4f" % (sys.argv[2], d["ms_per_step"]))
PY
}
for rep in 1 2 3; do
for tree in . _r5; do
  for cfg in ${CFGS:-fm}; do
  (cd $GRAFT_REPO_ROOT/$tree && timeout 300 python bench.py --config $cfg ${ARGS:---force-sharded} --steps ${STEPS:-60} --warmup 10 --no-cpu-baseline --no-extra-configs > $O/x.json 2> $O/x.err)
  line $O/x.json "tree $tree cfg $cfg rep $rep"
  done
done
done
