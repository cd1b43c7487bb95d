#!/bin/bash
# round 6: YoutubeDNN at 8 192 samples: with / without forks onto side streams inside the captured step (ops.config.fork_in_capture)
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r06aa
mkdir -p $O
export TMPDIR=/tmp
run() { # label, python prelude, bench args...
  label=$1; pre=$2; shift 2
  python -c "
import sys, runpy
import recbox_amd.ops as o
$pre
sys.argv = ['bench.py'] + '''$*'''.split()
runpy.run_path('bench.py', run_name='__main__')
" > $O/x.json 2> $O/x.err
  python - "$label" <<PY
import json, sys
try:
    d = json.loads([l for l in open("$O/x.json") if l.startswith("{")][-1])
    print(sys.argv[1], " ms_per_step %.4f" % d["ms_per_step"])
except Exception as e:
    print(sys.argv[1], "failed", e)
PY
}
for rep in 1 2 3; do
for b in 8192 65536; do
  A="--config youtubednn --batch $b --steps 40 --warmup 8 --no-cpu-baseline"
  run "forks on  B=$b rep $rep" "pass" $A
  run "forks off B=$b rep $rep" "o.config.fork_in_capture = False" $A
  run "dw_beside off B=$b rep $rep" "o.config.dw_beside_lookup = False" $A
done
done
